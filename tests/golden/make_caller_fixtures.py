#!/usr/bin/env python3
"""Run the reference's OWN Python caller against this build's binding shims (BUILD container only).

`upright_control/src/upright_control/{wrappers,manager}.py` are imported from /root/reference, unmodified, with
    sys.modules["upright_control.bindings"] = upright_amd.control_bindings
    sys.modules["upright_core.bindings"]    = upright_amd.core_bindings
i.e. exactly the substitution INTEGRATION.md section 2 describes.  What is absent from this image is stubbed as in
make_fixtures.py (spatialmath, rospkg, xacrodoc, mobile_manipulation_central, IPython) plus
    upright_control.robot                 (Pinocchio): build_robot_interfaces -> the serial chain of upright_amd/robots.py
    core.parsing.parse_and_compile_urdf   (xacro + ROS): returns the include names, ';'-joined (no file is compiled)

Outputs (tests/golden/), data only:
  settings_from_reference_wrappers.json   every field of the `ControllerSettings` object that the reference's
                                          wrappers.py:81-399 fills in, for the eleven golden configs
  manager_call_sequence.json              the calls the reference's manager.py:105-209 (`from_config`, `warmstart`, `step`, `plan`)
                                          makes on `bindings.ControllerInterface`, with their arguments, against a recording
                                          stand-in (tests/standin.py:RecordingControllerInterface), and what it returned per tick

Run:  python tests/golden/make_caller_fixtures.py      (after make_fixtures.py: reads configs.json's list of configs)
"""
import copy
import json
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(ROOT / "tests"))

import make_fixtures as mf  # noqa: E402

REF = mf.REF

CONFIGS = {
    "ur10_demo": "upright_cmd/config/demos/ur10_demo.yaml",
    "thing_demo": "upright_cmd/config/demos/thing_demo.yaml",
    "full_bottle_point1": "upright_cmd/config/ral23/experiments/freespace/full/full_bottle_point1.yaml",
    "full_arch_point3": "upright_cmd/config/ral23/experiments/freespace/full/full_arch_point3.yaml",
    "static_arch_point3": "upright_cmd/config/ral23/experiments/static_obstacles/full/full_arch_point3.yaml",
    "projectile_head_on": "upright_cmd/config/ral23/experiments/projectile/projectile_head_on.yaml",
    "robust_sim": "upright_robust/config/demos/sim.yaml",
    "sudden_t1.0": "upright_cmd/config/ral23/experiments/sudden_obstacle/sudden_t1.0.yaml",
    "full_cups_point1": "upright_cmd/config/ral23/experiments/freespace/full/full_cups_point1.yaml",
    "full_dice_point1": "upright_cmd/config/ral23/experiments/freespace/full/full_dice_point1.yaml",
    "full_bottle_arm_only": "upright_cmd/config/ral23/experiments/freespace/full/full_bottle_arm_only.yaml",
}
SEQUENCE_CONFIGS = ("thing_demo", "projectile_head_on", "ur10_demo")


def install():
    """The stubs of make_fixtures.py, then this build's shims in the two bindings slots and the reference's upright_control
    package on the path."""
    mf.install_stubs()
    if not hasattr(np, "infty"):
        np.infty = np.inf   # manager.py:114 uses the NumPy 1.x alias; this image has NumPy 2
    import upright_amd.control_bindings as cb
    import upright_amd.core_bindings as kb

    sys.modules["upright_core.bindings"] = kb
    sys.modules.pop("upright_control", None)   # (make_fixtures.py parks an empty module there for upright_robust)
    import upright_core as core   # the reference's package; its parsing.py now builds kb.RigidBody / kb.ContactPoint

    assert core.parsing.RigidBody is kb.RigidBody
    pkgs = {p: REF / p for p in ("upright_cmd", "upright_robust", "upright_assets", "upright_core")}
    core.parsing.parse_ros_path = lambda d, as_string=True: (pkgs[d["package"]] / d["path"]).as_posix() if as_string else pkgs[d["package"]] / d["path"]
    core.parsing.parse_and_compile_urdf = lambda d, **kw: ";".join(d.get("includes", []))

    # upright_control: the reference's package directory, with .bindings and .robot provided
    pkg = types.ModuleType("upright_control")
    pkg.__path__ = [str(REF / "upright_control" / "src" / "upright_control")]
    sys.modules["upright_control"] = pkg
    sys.modules["upright_control.bindings"] = cb
    pkg.bindings = cb
    robot = types.ModuleType("upright_control.robot")

    class _Robot:
        """What manager.py:140-150 needs of robot.py's PinocchioRobot: forward_xu + link_pose (position, xyzw quaternion)."""

        def __init__(self, settings):
            from upright_amd import robots

            self.nq = settings.dims.robot.q
            self.chain = robots.from_config({"base_type": cb.robot_base_type_to_string(settings.robot_base_type),
                                             "dims": {"q": self.nq}, "base_pose": list(settings.base_pose)})
            self.q = np.zeros(self.nq)

        def forward_xu(self, x, u=None):
            self.q = np.array(x[: self.nq], dtype=np.float64)

        def link_pose(self):
            from upright_amd.control import rot_to_quat_xyzw

            r, C = self.chain.forward(self.q)
            return r, rot_to_quat_xyzw(C)

    robot.build_robot_interfaces = lambda settings: (_Robot(settings), None)
    sys.modules["upright_control.robot"] = robot
    pkg.robot = robot
    import upright_control.manager as manager   # noqa: E402  (the reference's files, unmodified)
    import upright_control.wrappers as wrappers  # noqa: E402

    assert Path(wrappers.__file__).is_relative_to(REF) and Path(manager.__file__).is_relative_to(REF)
    return core, wrappers, manager, cb


def load_controller_config(core, key, rcfg_objects):
    d = core.parsing.load_config((REF / CONFIGS[key]).as_posix())
    c = d["controller"]
    if key == "robust_sim":   # what planning_sim_loop.py:513-534 writes into the config before parsing it (make_fixtures.py)
        c["objects"].update(rcfg_objects["objects"])
        c["arrangements"]["robust"] = rcfg_objects["arrangement"]
        c["balancing"]["arrangement"] = "robust"
        c["waypoints"] = [{"time": 0, "position": [-2.0, 1.0, 0], "orientation": [0, 0, 0, 1]}]
    return c


def robust_objects():
    h = 0.30
    names, objs = [], {}
    for i, (sx, sy, sz) in enumerate((a, b, c) for a in (-1, 1) for b in (-1, 1) for c in (-1, 1)):
        n = f"sim_block_{i + 1}"
        names.append(n)
        objs[n] = {"mass": 1.0, "shape": "cuboid", "side_lengths": [0.15, 0.15, h], "color": [1, 0, 0, 1],
                   "com_offset": [0.06 * sx, 0.06 * sy, 0.5 * h * sz]}
    arr = {"objects": [{"name": n, "type": n, "parent": "ee", "offset": {"x": 0}} for n in names],
           "contacts": [{"first": "ee", "second": n, "mu": 0.2, "support_area_inset": 0.0} for n in names]}
    return {"objects": objs, "arrangement": arr}


def main():
    core, wrappers, manager, cb = install()
    from standin import RecordingControllerInterface, dump_settings

    rob = robust_objects()
    settings = {}
    for key in CONFIGS:
        c = load_controller_config(core, key, rob)
        s = wrappers.ControllerSettings(copy.deepcopy(c))     # the reference's constructor, on this build's settings struct
        settings[key] = dump_settings(s)
    with open(HERE / "settings_from_reference_wrappers.json", "w") as f:
        json.dump(settings, f, indent=1, sort_keys=True)

    # --- the call sequence of manager.py against a recording ControllerInterface ----------------------------------------
    real = cb.ControllerInterface
    cb.ControllerInterface = RecordingControllerInterface
    seq = {}
    try:
        for key in SEQUENCE_CONFIGS:
            c = load_controller_config(core, key, rob)
            mgr = manager.ControllerManager.from_config(copy.deepcopy(c))
            rec = mgr.mpc
            mgr.warmstart()
            # a simulation loop in the style of upright_cmd/scripts/simulations/mpc_sim.py: the simulator's period (4 ms) is not
            # the controller's (tracking.min_policy_update_time = 10 ms); the observed state is the last policy state, perturbed
            outs = []
            x = np.array(mgr.model.settings.initial_state, dtype=np.float64)
            t = 0.0
            for i in range(50):
                xo, uo = mgr.step(t, x)
                outs.append([t, xo.tolist(), uo.tolist()])
                x = xo + 1e-3 * np.cos(np.arange(len(xo)) + i)
                t += 0.004
            ts, xs, us = mgr.get_mpc_trajectory()
            plan = mgr.plan(0.005, 0.1)
            seq[key] = {
                "timestep": mgr.timestep,
                "calls": rec.calls,
                "step_outputs": outs,
                "replanning_times": list(mgr.replanning_times),
                "last_planning_time": mgr.last_planning_time,
                "trajectory_shapes": [list(np.shape(ts)), list(np.shape(xs)), list(np.shape(us))],
                "plan": {"ts": np.asarray(plan.ts).tolist(), "xs": np.asarray(plan.xs).tolist(), "us": np.asarray(plan.us).tolist()},
            }
    finally:
        cb.ControllerInterface = real
    with open(HERE / "manager_call_sequence.json", "w") as f:
        json.dump(seq, f, separators=(",", ":"))
    print("settings:", {k: len(json.dumps(v)) for k, v in settings.items()})
    print("sequences:", {k: len(v["calls"]) for k, v in seq.items()})


if __name__ == "__main__":
    main()
