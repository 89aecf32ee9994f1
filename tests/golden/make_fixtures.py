#!/usr/bin/env python3
"""Generate golden fixtures by importing the reference's Python (BUILD container only).

Follows SURVEY.md Appendix A.  Imports `/root/reference/upright_core/src/upright_core`
({math,polyhedron,parsing}.py) behind small stubs for the packages that are absent here
(spatialmath, rospkg, xacrodoc, mobile_manipulation_central, IPython, upright_core.bindings),
runs `parse_control_objects` / `load_config` on the reference's own YAML files and dumps the
numeric results as JSON.  Only the JSON travels; the reference sources never do.

Outputs (tests/golden/):
  arrangements.json   bodies + contacts for the arrangements the hot path is benchmarked/tested on
                      (from upright_cmd/config/{controller,arrangements}.yaml and
                      upright_core/tests/config.yaml)
  configs.json        merged controller dicts (numeric fields the ControllerSettings mirror reads)
                      for the BASELINE.json configs, via the reference's include resolver
  parse_dsl.json      parse_number / parse_array known answers
  inertial.json       upright_robust.modelling.UncertainObject.M (spatial mass matrix about the EE origin) of every body of three
                      arrangements and upright_robust.utils.body_gravity6 at three orientations: the reference's own numpy
                      statement of the inertial half of the object-dynamics residual (third reference-held answer for a1 / a3)
  grasp.json          upright_robust.modelling.compute_grasp_matrix (contact forces -> body wrenches) for three arrangements:
                      a statement of the wrench map that is independent of the C++ (second reference-held answer for a1 / a2)

Run:  python tests/golden/make_fixtures.py
"""
import copy
import json
import sys
import types
from pathlib import Path

import numpy as np
import yaml

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


# ---------------------------------------------------------------------------------------------
# stubs (Appendix A step 2)
def _q2r(q, order="sxyz"):
    q = np.asarray(q, dtype=float)
    if order == "xyzs":
        x, y, z, s = q
    else:
        s, x, y, z = q
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - s * z), 2 * (x * z + s * y)],
            [2 * (x * y + s * z), 1 - 2 * (x * x + z * z), 2 * (y * z - s * x)],
            [2 * (x * z - s * y), 2 * (y * z + s * x), 1 - 2 * (x * x + y * y)],
        ]
    )


def _r2q(R, order="sxyz"):
    R = np.asarray(R, dtype=float)
    tr = np.trace(R)
    s = 0.5 * np.sqrt(max(0.0, 1.0 + tr))
    kx = R[2, 1] - R[1, 2]
    ky = R[0, 2] - R[2, 0]
    kz = R[1, 0] - R[0, 1]
    # largest-diagonal branch for robustness
    d = np.array([R[0, 0], R[1, 1], R[2, 2]])
    i = int(np.argmax(d))
    if i == 0:
        kx1 = R[0, 0] - R[1, 1] - R[2, 2] + 1
        ky1 = R[1, 0] + R[0, 1]
        kz1 = R[2, 0] + R[0, 2]
        sgn = kx >= 0
    elif i == 1:
        kx1 = R[1, 0] + R[0, 1]
        ky1 = R[1, 1] - R[0, 0] - R[2, 2] + 1
        kz1 = R[2, 1] + R[1, 2]
        sgn = ky >= 0
    else:
        kx1 = R[2, 0] + R[0, 2]
        ky1 = R[2, 1] + R[1, 2]
        kz1 = R[2, 2] - R[0, 0] - R[1, 1] + 1
        sgn = kz >= 0
    if sgn:
        kx += kx1
        ky += ky1
        kz += kz1
    else:
        kx -= kx1
        ky -= ky1
        kz -= kz1
    nm = np.linalg.norm([kx, ky, kz])
    if nm == 0:
        v = np.zeros(3)
        s = 1.0
    else:
        v = np.sqrt(max(0.0, 1 - s * s)) / nm * np.array([kx, ky, kz])
    if order == "xyzs":
        return np.array([v[0], v[1], v[2], s])
    return np.array([s, v[0], v[1], v[2]])


def _qunit(q):
    q = np.asarray(q, dtype=float)
    return q / np.linalg.norm(q)


def _rotx(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def _roty(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _rotz(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


class _RigidBody:
    def __init__(self, mass, inertia, com):
        self.mass = mass
        self.inertia = np.array(inertia, dtype=float)
        self.com = np.array(com, dtype=float)


class _ContactPoint:
    pass


def install_stubs():
    sm = types.ModuleType("spatialmath")
    smb = types.ModuleType("spatialmath.base")
    for name, fn in dict(q2r=_q2r, r2q=_r2q, qunit=_qunit, rotx=_rotx, roty=_roty, rotz=_rotz).items():
        setattr(smb, name, fn)
    sm.base = smb
    sys.modules["spatialmath"] = sm
    sys.modules["spatialmath.base"] = smb
    for name in ("rospkg", "mobile_manipulation_central", "IPython"):
        sys.modules[name] = types.ModuleType(name)
    xd = types.ModuleType("xacrodoc")
    xd.XacroDoc = object
    sys.modules["xacrodoc"] = xd
    sys.path.insert(0, str(REF / "upright_core" / "src"))
    b = types.ModuleType("upright_core.bindings")
    b.RigidBody = _RigidBody
    b.ContactPoint = _ContactPoint
    sys.modules["upright_core.bindings"] = b
    # upright_robust.modelling (grasp matrix): rigeo is touched inside methods the fixtures never call, upright_control only by
    # upright_robust.parsing (imported by the package's __init__, not used here)
    for name in ("rigeo", "upright_control"):
        sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, str(REF / "upright_robust" / "src"))


# ---------------------------------------------------------------------------------------------
def dump_arrangement(core, cfg, name):
    cfg = copy.deepcopy(cfg)
    cfg.setdefault("balancing", {})["arrangement"] = name
    bodies, contacts = core.parsing.parse_control_objects(cfg)
    out = {"bodies": [], "contacts": []}
    for bname in sorted(bodies):  # std::map order (contact_constraints.h:180)
        b = bodies[bname]
        I = np.asarray(b.inertia)
        out["bodies"].append(
            {
                "name": bname,
                "mass": float(b.mass),
                "com": np.asarray(b.com).tolist(),
                "inertia": I.tolist(),
                # rigid_body.h:47-51  [m, m*c, vech(I)]
                "params": [float(b.mass)]
                + (b.mass * np.asarray(b.com)).tolist()
                + [I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]],
            }
        )
    for c in contacts:
        out["contacts"].append(
            {
                "object1_name": c.object1_name,
                "object2_name": c.object2_name,
                "mu": float(c.mu),
                "normal": np.asarray(c.normal).tolist(),
                "span": np.asarray(c.span).tolist(),
                "r_co_o1": np.asarray(c.r_co_o1).tolist(),
                "r_co_o2": np.asarray(c.r_co_o2).tolist(),
            }
        )
    return out


def arrangement_input(cfg, name):
    """The INPUT side of one arrangement: the arrangement entry and the object-type entries it uses (plain data
    out of the reference's YAML files), so that the arrangement parser of the build can be checked against
    `dump_arrangement`'s output without reading the reference tree."""
    arr = copy.deepcopy(cfg["arrangements"][name])
    types = {"ee"} | {o["type"] for o in arr["objects"]}
    return {"arrangement": arr, "objects": {t: copy.deepcopy(cfg["objects"][t]) for t in sorted(types)}}


def jsonable(x):
    if isinstance(x, dict):
        return {str(k): jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (np.floating, np.integer)):
        return x.item()
    return x


def main():
    install_stubs()
    import upright_core as core

    # include resolver: package -> directory map instead of rospkg (Appendix A)
    pkgs = {p: REF / p for p in ("upright_cmd", "upright_robust", "upright_assets", "upright_core")}

    def parse_ros_path(d, as_string=True):
        p = pkgs[d["package"]] / d["path"]
        return p.as_posix() if as_string else p

    core.parsing.parse_ros_path = parse_ros_path

    # --- arrangements ------------------------------------------------------------------------
    with open(REF / "upright_cmd/config/controller.yaml") as f:
        ctrl = yaml.safe_load(f)
    ctrl.pop("include")
    with open(REF / "upright_cmd/config/arrangements.yaml") as f:
        arr = yaml.safe_load(f)
    cfg = core.parsing.recursive_dict_update(copy.deepcopy(arr), ctrl)
    arrangements = {}
    inputs = {}
    for name in ("pink_bottle", "foam_die2", "box_arch", "blue_cups", "wedge", "simulation_box_with_fixture"):
        inputs[name] = arrangement_input(cfg, name)
        arrangements[name] = dump_arrangement(core, cfg, name)

    with open(REF / "upright_core/tests/config.yaml") as f:
        tcfg = yaml.safe_load(f)
    for name in ("box", "cylinder_box", "wedge_box"):
        inputs["tests/" + name] = arrangement_input(tcfg, name)
        arrangements["tests/" + name] = dump_arrangement(core, tcfg, name)

    # BASELINE config 4 (upright_robust): the arrangement planning_sim_loop.py:454-534 assembles at run time --
    # one 0.15 x 0.15 x h cuboid of mass 1 per vertex of the CoM box (+-0.06, +-0.06, +-h/2), all at the same place
    # on the tray, mu = 0.2, no support-area inset.  The dicts below carry the same VALUES that script writes
    # into the config (h = 0.30 m); bodies / contacts come out of the reference's own parser.
    h = 0.30
    rcfg = copy.deepcopy(cfg)
    names = []
    for i, (sx, sy, sz) in enumerate((a, b, c) for a in (-1, 1) for b in (-1, 1) for c in (-1, 1)):
        n = f"sim_block_{i + 1}"
        names.append(n)
        rcfg["objects"][n] = {"mass": 1.0, "shape": "cuboid", "side_lengths": [0.15, 0.15, h], "color": [1, 0, 0, 1],
                              "com_offset": [0.06 * sx, 0.06 * sy, 0.5 * h * sz]}
    rcfg["arrangements"]["robust_8corner"] = {
        "objects": [{"name": n, "type": n, "parent": "ee", "offset": {"x": 0}} for n in names],
        "contacts": [{"first": "ee", "second": n, "mu": 0.2, "support_area_inset": 0.0} for n in names],
    }
    inputs["robust_8corner"] = arrangement_input(rcfg, "robust_8corner")
    arrangements["robust_8corner"] = dump_arrangement(core, rcfg, "robust_8corner")
    with open(OUT / "arrangement_inputs.json", "w") as f:
        json.dump(jsonable(inputs), f, indent=1)

    with open(OUT / "arrangements.json", "w") as f:
        json.dump(arrangements, f, indent=1)

    # --- grasp matrices (upright_robust/modelling.py:83-103): contact forces -> body wrenches about the EE origin ----------
    import upright_robust.modelling as mdl

    grasp = {}
    for name, c in (("pink_bottle", cfg), ("box_arch", cfg), ("robust_8corner", rcfg)):
        c2 = copy.deepcopy(c)
        c2.setdefault("balancing", {})["arrangement"] = name
        bodies, contacts = core.parsing.parse_control_objects(c2)
        names = sorted(bodies)            # the body order of the stacked residual (contact_constraints.h:179-192)
        index = mdl.compute_object_name_index(names)
        G = mdl.compute_grasp_matrix(index, [mdl.RobustContactPoint(cp) for cp in contacts])
        # ... and the span form of every contact's friction cone (modelling.py:34-44): the four generators normal +- mu span_i.  The
        # cone they span, {f_n >= 0, |t_0| + |t_1| <= mu f_n}, is the linearised cone of contact_constraints.h:50-77 -- a
        # reference-held answer for the friction rows (a4)
        grasp[name] = {"names": names, "G": G.tolist(), "S": [mdl.RobustContactPoint(cp).S.tolist() for cp in contacts],
                       "mu": [float(cp.mu) for cp in contacts]}
    with open(OUT / "grasp.json", "w") as f:
        json.dump(grasp, f, indent=1)

    # --- spatial mass matrices and the gravity twist (upright_robust/modelling.py:47-77, utils.py:5-13) -----------------------
    import upright_robust.utils as rutils

    inertial = {"arrangements": {}, "gravity": []}
    for name, c in (("pink_bottle", cfg), ("box_arch", cfg), ("robust_8corner", rcfg)):
        c2 = copy.deepcopy(c)
        c2.setdefault("balancing", {})["arrangement"] = name
        bodies, _ = core.parsing.parse_control_objects(c2)
        inertial["arrangements"][name] = {
            "names": sorted(bodies),
            "M": [mdl.UncertainObject(bodies[n]).M.tolist() for n in sorted(bodies)],
        }
    for C_ew in (np.eye(3), _rotx(0.3) @ _roty(-0.2), _rotz(1.1) @ _roty(0.4) @ _rotx(-0.7)):
        inertial["gravity"].append({"C_ew": C_ew.tolist(), "G": np.asarray(rutils.body_gravity6(C_ew)).tolist()})
    with open(OUT / "inertial.json", "w") as f:
        json.dump(inertial, f, indent=1)

    # --- merged configs for the BASELINE configs ------------------------------------------------
    configs = {}
    for key, rel in {
        "ur10_demo": "upright_cmd/config/demos/ur10_demo.yaml",
        "thing_demo": "upright_cmd/config/demos/thing_demo.yaml",
        "full_bottle_point1": "upright_cmd/config/ral23/experiments/freespace/full/full_bottle_point1.yaml",
        "full_arch_point3": "upright_cmd/config/ral23/experiments/freespace/full/full_arch_point3.yaml",
        # BASELINE config 3: three stacked objects + the static obstacles of obstacles/simple.yaml
        "static_arch_point3": "upright_cmd/config/ral23/experiments/static_obstacles/full/full_arch_point3.yaml",
        # BASELINE config 5: dynamic obstacle (obstacles/dynamic.yaml) + projectile-path constraint
        "projectile_head_on": "upright_cmd/config/ral23/experiments/projectile/projectile_head_on.yaml",
        # BASELINE config 4: the upright_robust planning demo (slacks, init_sqp_iteration 3, T = 10 s); its arrangement
        # is assembled at run time by planning_sim_loop.py:454-534 (robust_8corner above carries the same values)
        "robust_sim": "upright_robust/config/demos/sim.yaml",
        # the sudden-obstacle experiments of the paper (not a BASELINE config; the same path): static obstacles of simple.yaml + ONE
        # dynamic obstacle whose position jumps at t = 1 s (two `modes`; the simulator / Vicon re-sets the observed state)
        "sudden_t1.0": "upright_cmd/config/ral23/experiments/sudden_obstacle/sudden_t1.0.yaml",
        # the paper's other free-space arrangements (VERDICT r03 missing 1): seven cups (star, with friction), two stacked dice,
        # and the bottle on the arm alone (base joints locked)
        "full_cups_point1": "upright_cmd/config/ral23/experiments/freespace/full/full_cups_point1.yaml",
        "full_dice_point1": "upright_cmd/config/ral23/experiments/freespace/full/full_dice_point1.yaml",
        "full_bottle_arm_only": "upright_cmd/config/ral23/experiments/freespace/full/full_bottle_arm_only.yaml",
    }.items():
        d = core.parsing.load_config((REF / rel).as_posix())
        c = d["controller"]
        if key == "robust_sim":   # what planning_sim_loop.py:513-534 writes into the config before parsing it
            c["objects"].update({n: rcfg["objects"][n] for n in names})
            c["arrangements"]["robust"] = rcfg["arrangements"]["robust_8corner"]
            c["balancing"]["arrangement"] = "robust"
            c["waypoints"] = [{"time": 0, "position": [-2.0, 1.0, 0], "orientation": [0, 0, 0, 1]}]
        # keep what the ControllerSettings mirror reads; drop the (large) unrelated object tables
        keep = {
            k: c[k]
            for k in (
                "gravity", "mpc", "rollout", "sqp", "balancing", "tracking", "estimation", "weights",
                "limits", "waypoints", "obstacles", "inertial_alignment", "end_effector_box_constraint",
                "projectile_path_constraint", "operating_points", "debug", "recompile_libraries",
            )
            if k in c
        }
        keep["robot"] = {k: v for k, v in c["robot"].items() if k != "urdf"}
        keep["objects"] = c["objects"]
        keep["arrangements"] = {c["balancing"]["arrangement"]: c["arrangements"][c["balancing"]["arrangement"]]}
        # parsed numerics (what wrappers.py:81-399 computes from the dict)
        parsed = {
            "x0": core.parsing.parse_array(c["robot"]["x0"]),
            "input_weight": core.parsing.parse_diag_matrix_dict(c["weights"]["input"]),
            "state_weight": core.parsing.parse_diag_matrix_dict(c["weights"]["state"]),
            "end_effector_weight": core.parsing.parse_diag_matrix_dict(c["weights"]["end_effector"]),
            "input_limit_lower": core.parsing.parse_array(c["limits"]["input"]["lower"]),
            "input_limit_upper": core.parsing.parse_array(c["limits"]["input"]["upper"]),
            "state_limit_lower": core.parsing.parse_array(c["limits"]["state"]["lower"]),
            "state_limit_upper": core.parsing.parse_array(c["limits"]["state"]["upper"]),
            "time_horizon": core.parsing.parse_number(c["mpc"]["time_horizon"]),
            "dt": core.parsing.parse_number(c["sqp"]["dt"]),
        }
        bodies, contacts = core.parsing.parse_control_objects(copy.deepcopy(c))
        parsed["n_bodies"] = len(bodies)
        parsed["n_contacts"] = len(contacts)
        configs[key] = {"controller": jsonable(keep), "parsed": jsonable(parsed)}
    with open(OUT / "configs.json", "w") as f:
        json.dump(configs, f, indent=1)

    # --- number / array DSL ------------------------------------------------------------------------
    dsl = {
        "numbers": [[s, core.parsing.parse_number(s)] for s in ("2pi", "0.5pi", "-0.25pi", "1e-3", 3, "0.417pi")],
        "arrays": [
            [a, core.parsing.parse_array(a).tolist()]
            for a in (
                ["0rep3", "1", "2pi"],
                ["-1", "1", "0", "0.5pi", "-0.25pi", "0.5pi", "-0.25pi", "0.5pi", "0.417pi", "0rep9", "0rep9"],
                ["1rep9"],
            )
        ],
    }
    with open(OUT / "parse_dsl.json", "w") as f:
        json.dump(jsonable(dsl), f, indent=1)

    # --- reference's own unit tests (Appendix A step 5): run them against the stubs ---------------
    print("arrangements:", {k: (len(v["bodies"]), len(v["contacts"])) for k, v in arrangements.items()})
    print("configs:", list(configs))


if __name__ == "__main__":
    main()
