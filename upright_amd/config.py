"""Configuration front-end: YAML with recursive `include:` and the number / array mini-DSL.

Own implementation of the semantics of `upright_core/src/upright_core/parsing.py:14-106` and
docs/configuration.md:6-44 (includes are loaded first, in order; the including file overrides; `key`
nests an include; maximum inclusion depth 5).  ROS package paths are resolved through an explicit
{package: directory} map instead of rospkg.  Checked against the reference's own outputs in
tests/golden/{configs,parse_dsl}.json.

Arrangement -> (bodies, contact points) parsing (`parsing.py:351-410`, `polyhedron.py`) is the next
row of SURVEY.md section 8f and is not here yet: bodies / contacts are supplied by the caller or taken from
tests/golden/arrangements.json (generated from the reference).
"""
from pathlib import Path

import numpy as np
import yaml

PACKAGE_DIRS = {}


def register_package(name, directory):
    PACKAGE_DIRS[name] = Path(directory)


def recursive_dict_update(default, custom):
    if not isinstance(default, dict) or not isinstance(custom, dict):
        raise TypeError("Params of recursive_update should be dicts")
    for key, val in custom.items():
        if isinstance(val, dict) and isinstance(default.get(key), dict):
            default[key] = recursive_dict_update(default[key], val)
        else:
            default[key] = val
    return default


def parse_ros_path(d, as_string=True):
    try:
        base = PACKAGE_DIRS[d["package"]]
    except KeyError:
        raise KeyError(f"package '{d['package']}' is not registered (upright_amd.config.register_package)")
    p = base / d["path"]
    return p.as_posix() if as_string else p


def load_config(path, depth=0, max_depth=5):
    if depth > max_depth:
        raise Exception(f"Maximum inclusion depth {max_depth} exceeded.")
    with open(path) as f:
        d = yaml.safe_load(f)
    includes = d.pop("include", [])
    merged = {}
    for inc in includes:
        sub = load_config(parse_ros_path(inc), depth=depth + 1, max_depth=max_depth)
        if "key" in inc:
            sub = {inc["key"]: sub}
        merged = recursive_dict_update(merged, sub)
    return recursive_dict_update(merged, d)


def parse_number(x, dtype=float):
    """'2pi' -> 2 * pi; anything else through dtype (parsing.py:63-71)."""
    if type(x) == str and x.endswith("pi"):
        return dtype(x[:-2]) * np.pi
    return dtype(x)


def _parse_array_element(x):
    try:
        return [float(x)]
    except ValueError:
        if x.endswith("pi"):
            return [float(x[:-2]) * np.pi]
        if "rep" in x:
            y, n = x.split("rep")
            return float(y) * np.ones(int(n))
        raise ValueError(f"Could not convert {x} to array element.")


def parse_array(a):
    """['0rep3', '1', '2pi'] -> [0, 0, 0, 1, 6.283...] (parsing.py:74-91)."""
    return np.concatenate([_parse_array_element(x) for x in a])


def parse_diag_matrix_dict(d):
    return parse_number(d["scale"]) * np.diag(parse_array(d["diag"]))


def parse_support_offset(d):
    x = d.get("x", 0)
    y = d.get("y", 0)
    if "r" in d and "θ" in d:
        r, th = d["r"], parse_number(d["θ"])
        x += r * np.cos(th)
        y += r * np.sin(th)
    elif "r" in d or "θ" in d:
        raise ValueError("Radius and angle must *both* be specified in support offset.")
    return np.array([x, y])
