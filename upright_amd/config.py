"""Configuration front-end: YAML files layered through `include:` and the number / array mini-language.

Semantics follow docs/configuration.md:6-44 of the reference and are pinned by its own outputs
(tests/golden/{configs,parse_dsl}.json, produced by importing the reference's `load_config` / `parse_*`,
`upright_core/src/upright_core/parsing.py:30-106`):

* a file's `include` entries are resolved first, in list order, each to a complete layer of its own;
  `key` nests a layer; later layers override earlier ones and the including file overrides them all;
  mappings merge key by key, anything else is replaced; inclusion nests at most five levels deep;
* a scalar may be written "<c>pi" (c times pi); an array is a list whose items are numbers, "<c>pi", or
  "<v>rep<n>" (v repeated n times).

ROS package names are resolved through an explicit {package: directory} registry (`register_package`)
instead of rospkg.  The arrangement -> (bodies, contact points) step lives in `upright_amd/arrangement.py`.
"""
import math
import re
from pathlib import Path

import numpy as np
import yaml

MAX_INCLUDE_DEPTH = 5
_PACKAGES = {}

# "<c>pi" and "<v>rep<n>": the coefficient / value is any float literal, the count a non-negative integer
_FLOAT = r"[+-]?(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?"
_PI_TERM = re.compile(rf"^\s*({_FLOAT})pi\s*$")
_REP_TERM = re.compile(rf"^\s*({_FLOAT})rep(\d+)\s*$")


def register_package(name, directory):
    """Tell the loader where the files of ROS package `name` live."""
    _PACKAGES[name] = Path(directory)


def package_path(entry, as_string=True):
    """{package, path} -> file path (the reference's `parse_ros_path`)."""
    name = entry["package"]
    if name not in _PACKAGES:
        raise KeyError(f"package '{name}' is not registered (upright_amd.config.register_package)")
    target = _PACKAGES[name] / entry["path"]
    return target.as_posix() if as_string else target


parse_ros_path = package_path   # the reference's name for it


def overlay(base, top):
    """Merge mapping `top` onto mapping `base` in place and return `base`: nested mappings merge, every other
    value of `top` replaces the one below."""
    if not (isinstance(base, dict) and isinstance(top, dict)):
        raise TypeError("overlay() merges two mappings")
    pending = [(base, top)]
    while pending:
        lower, upper = pending.pop()
        for key, value in upper.items():
            below = lower.get(key)
            if isinstance(value, dict) and isinstance(below, dict):
                pending.append((below, value))
            else:
                lower[key] = value
    return base


recursive_dict_update = overlay   # the reference's name for it


def _layers(path, depth, max_depth):
    """The mappings that make up the file at `path`, lowest priority first."""
    if depth > max_depth:
        raise Exception(f"Maximum inclusion depth {max_depth} exceeded.")
    with open(path) as stream:
        own = yaml.safe_load(stream) or {}
    stack = []
    for entry in own.pop("include", None) or ():
        layer = {}
        for part in _layers(package_path(entry), depth + 1, max_depth):
            overlay(layer, part)
        stack.append({entry["key"]: layer} if "key" in entry else layer)
    stack.append(own)
    return stack


def load_config(path, depth=0, max_depth=MAX_INCLUDE_DEPTH):
    """Read a YAML file and fold in everything it includes."""
    merged = {}
    for layer in _layers(path, depth, max_depth):
        overlay(merged, layer)
    return merged


def parse_number(x, dtype=float):
    """'2pi' -> 2 pi; everything else through `dtype`."""
    if isinstance(x, str):
        m = _PI_TERM.match(x)
        if m:
            return dtype(m.group(1)) * math.pi
    return dtype(x)


def _expand(item):
    """One array item -> the list of floats it stands for."""
    if not isinstance(item, str):
        return [float(item)]
    m = _REP_TERM.match(item)
    if m:
        return [float(m.group(1))] * int(m.group(2))
    m = _PI_TERM.match(item)
    if m:
        return [float(m.group(1)) * math.pi]
    try:
        return [float(item)]
    except ValueError:
        raise ValueError(f"Could not convert {item} to array element.") from None


def parse_array(items):
    """['0rep3', '1', '2pi'] -> array([0, 0, 0, 1, 6.283...])."""
    out = []
    for item in items:
        out.extend(_expand(item))
    return np.array(out, dtype=np.float64)


def parse_diag_matrix_dict(spec):
    """{scale, diag} -> scale * diag(array)."""
    return parse_number(spec["scale"]) * np.diag(parse_array(spec["diag"]))


def parse_support_offset(spec):
    """Planar offset of an object on its support: Cartesian part (x, y) plus an optional polar part (r, θ)."""
    has_r, has_angle = "r" in spec, "θ" in spec
    if has_r != has_angle:
        raise ValueError("Radius and angle must *both* be specified in support offset.")
    offset = np.array([spec.get("x", 0), spec.get("y", 0)], dtype=np.float64)
    if has_r:
        angle = parse_number(spec["θ"])
        offset += spec["r"] * np.array([math.cos(angle), math.sin(angle)])
    return offset
