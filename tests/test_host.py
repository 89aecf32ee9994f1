"""CPU tests of the host side: config front-end against the golden outputs of the reference's parser, the
bindings mirror, and the C-ABI library (loads, exports every declared symbol; no compute without a GPU)."""
import ctypes as C
import json
import re
from pathlib import Path

import numpy as np
import pytest

from upright_amd import _capi, config, control, control_bindings, core_bindings, robots
from upright_amd.distributed import shard_range

ROOT = Path(__file__).resolve().parents[1]
GOLD = ROOT / "tests" / "golden"


def test_number_and_array_dsl_against_reference_outputs():
    g = json.load(open(GOLD / "parse_dsl.json"))
    for s, v in g["numbers"]:
        assert np.isclose(config.parse_number(s), v, rtol=0, atol=0)
    for a, v in g["arrays"]:
        assert np.array_equal(config.parse_array(a), np.array(v))
    with pytest.raises(ValueError):
        config.parse_array(["abc"])
    with pytest.raises(ValueError):
        config.parse_support_offset({"r": 1})
    assert np.allclose(config.parse_support_offset({"x": 1, "y": 2, "r": 1, "θ": "0.5pi"}), [1, 3])


def test_include_resolution(tmp_path):
    """docs/configuration.md:6-44: includes first and in order, includer overrides, `key` nests, depth cap."""
    (tmp_path / "pkg").mkdir()
    (tmp_path / "pkg" / "base.yaml").write_text("a: 1\nb: {c: 2, d: 3}\n")
    (tmp_path / "pkg" / "mid.yaml").write_text("include:\n  - {package: pkg, path: base.yaml}\nb: {c: 20}\ne: 5\n")
    (tmp_path / "pkg" / "top.yaml").write_text(
        "include:\n  - {package: pkg, path: mid.yaml}\n  - {key: sub, package: pkg, path: base.yaml}\nb: {d: 30}\n")
    config.register_package("pkg", tmp_path / "pkg")
    d = config.load_config(tmp_path / "pkg" / "top.yaml")
    assert d == {"a": 1, "b": {"c": 20, "d": 30}, "e": 5, "sub": {"a": 1, "b": {"c": 2, "d": 3}}}
    (tmp_path / "pkg" / "loop.yaml").write_text("include:\n  - {package: pkg, path: loop.yaml}\n")
    with pytest.raises(Exception, match="inclusion depth"):
        config.load_config(tmp_path / "pkg" / "loop.yaml")


@pytest.mark.parametrize("name,arr", [("full_bottle_point1", "pink_bottle"), ("thing_demo", "pink_bottle"), ("ur10_demo", "pink_bottle"),
                                      ("sudden_t1.0", "pink_bottle"), ("projectile_head_on", "pink_bottle")])
def test_controller_settings_match_reference_parse(arrangements, name, arr):
    """Row P of SURVEY.md section 8a: the merged controller dict of the reference (golden) through our
    ControllerSettings gives field-for-field the numbers the reference's wrappers.py computes."""
    g = json.load(open(GOLD / "configs.json"))[name]
    c, par = g["controller"], g["parsed"]
    bodies, contacts = control.objects_from_fixture(arrangements[arr])
    s = control.ControllerSettings(c, bodies=bodies, contacts=contacts)
    assert np.array_equal(s.initial_state[:len(par["x0"])], np.array(par["x0"]))   # (the robot part: obstacle states follow, below)
    for k in ("input_weight", "state_weight", "end_effector_weight", "input_limit_lower", "input_limit_upper",
              "state_limit_lower", "state_limit_upper"):
        assert np.array_equal(getattr(s, k), np.array(par[k])), k
    assert s.mpc.time_horizon == par["time_horizon"] and s.sqp.dt == par["dt"]
    assert s.dims.c == par["n_contacts"] and len(s.balancing_settings.bodies) == par["n_bodies"]
    assert s.dims.nf == (1 if c["balancing"]["frictionless"] else 3)
    assert s.dims.x() == c["robot"]["dims"]["x"] + 9 * s.dims.o and s.dims.u() == c["robot"]["dims"]["u"] + s.dims.nf * s.dims.c   # dimensions.h:32-45
    assert s.sqp.hpipm.iter_max == 30 and s.sqp.sqp_iteration == 1 and s.tracking.min_policy_update_time == 0.01
    # dynamic obstacles (wrappers.py:360-396): modes parsed in order, the interface state seeded with modes[0] at rest
    dyn = (c.get("obstacles") or {}).get("dynamic") or []
    if c.get("obstacles", {}).get("enabled") and dyn:
        obs = list(s.obstacle_settings.dynamic_obstacles)
        assert len(obs) == len(dyn) and s.dims.o == len(dyn) and len(s.initial_state) == s.dims.robot.x + 9 * len(dyn)
        for o, oc in zip(obs, dyn):
            assert o.name == oc["name"] and o.radius == oc["radius"] and len(o.modes) == len(oc["modes"])
            for mo, mc in zip(o.modes, oc["modes"]):
                assert mo.time == mc["time"] and np.array_equal(mo.position, mc["position"]) and np.array_equal(mo.velocity, mc["velocity"])


def test_problem_from_settings_and_unsupported_terms(arrangements):
    g = json.load(open(GOLD / "configs.json"))["full_bottle_point1"]["controller"]
    bodies, contacts = control.objects_from_fixture(arrangements["pink_bottle"])
    s = control.ControllerSettings(g, bodies=bodies, contacts=contacts)
    P = control_bindings.problem_from_settings(s)
    assert (P.nx, P.nu, P.N, P.nf, P.nb, P.nc) == (27, 21, 20, 3, 1, 4)
    assert np.all(P.u_lb[9:] == -100) and np.all(P.u_ub[9:] == 100) and np.all(P.Rdiag[9:] == 0.001)
    assert np.allclose(P.body_params[0], arrangements["pink_bottle"]["bodies"][0]["params"])
    # obstacle avoidance: named sphere pairs (obstacles/simple.yaml:11-41) become the collision model
    s.obstacle_settings.enabled = True
    s.obstacle_settings.minimum_distance = 0.1
    for pair in (("wrist1_collision_link_0", "sphere1_top_link_0"), ("base_collision_link_0", "sphere2_bottom_link_0"),
                 ("wrist1_collision_link_0", "shoulder_collision_link_0")):
        s.obstacle_settings.collision_link_pairs.push_back(pair)
    Po = control_bindings.problem_from_settings(s)
    assert len(Po.pair_a) == 3 and len(Po.sph_r) == 5 and Po.obs_min_dist == 0.1
    assert list(Po.sph_frame) == [6, -1, 2, -1, 4]        # wrist_1 link, world, base link, world, upper arm link
    assert np.allclose(Po.sph_off[1], [0, 0.25, 0.75]) and np.allclose(Po.sph_r, [0.15, 0.25, 0.5, 0.25, 0.15])
    s.obstacle_settings.collision_link_pairs.push_back(("no_such_link_0", "sphere1_top_link_0"))
    with pytest.raises(RuntimeError, match="unknown collision object"):
        control_bindings.problem_from_settings(s)
    s.obstacle_settings.collision_link_pairs.pop()
    s.obstacle_settings.enabled = False
    s.balancing_settings.enabled = False
    with pytest.raises(RuntimeError):
        control_bindings.problem_from_settings(s)
    with pytest.raises(RuntimeError):
        control_bindings.robot_base_type_from_string("hovering")
    # without bodies= / contacts= the arrangement named in the config is parsed (wrappers.py:303-305)
    s2 = control.ControllerSettings(g)
    P2 = control_bindings.problem_from_settings(s2)
    assert np.allclose(P2.body_params, P.body_params, atol=1e-12) and np.allclose(P2.contact_r2, P.contact_r2, atol=1e-12)
    assert np.array_equal(P2.contact_mu, P.contact_mu) and np.allclose(P2.contact_span, P.contact_span, atol=1e-12)


@pytest.mark.parametrize("name", ["pink_bottle", "foam_die2", "box_arch", "blue_cups", "wedge", "simulation_box_with_fixture",
                                  "tests/box", "tests/cylinder_box", "tests/wedge_box", "robust_8corner"])
def test_arrangement_parser_matches_reference_outputs(arrangements, name):
    """the arrangement front-end (solids stacked on the tray -> bodies, contact polygons, insets)
    against what the reference's parse_control_objects produced from the same YAML entries.  Contact ORDER matters:
    it is the layout of the force block of u.  1e-9: the reference finds stacking heights with a linear programme."""
    import copy

    from upright_amd.arrangement import parse_control_objects

    inp = json.load(open(GOLD / "arrangement_inputs.json"))[name]
    cfg = {"objects": copy.deepcopy(inp["objects"]), "arrangements": {name: copy.deepcopy(inp["arrangement"])}, "balancing": {"arrangement": name}}
    bodies, contacts = parse_control_objects(cfg)
    exp = arrangements[name]
    eb = {b["name"]: b for b in exp["bodies"]}
    assert set(bodies) == set(eb) and len(contacts) == len(exp["contacts"])
    for n, b in bodies.items():
        assert abs(b.mass - eb[n]["mass"]) < 1e-12
        assert np.abs(b.com - np.array(eb[n]["com"])).max() < 1e-9
        assert np.abs(b.inertia - np.array(eb[n]["inertia"])).max() < 1e-12
        assert np.abs(b.get_parameters() - np.array(eb[n]["params"])).max() < 1e-9
    for c, e in zip(contacts, exp["contacts"]):
        assert (c.object1_name, c.object2_name) == (e["object1_name"], e["object2_name"])
        assert abs(c.mu - e["mu"]) < 1e-15
        for k in ("normal", "span", "r_co_o1", "r_co_o2"):
            assert np.abs(np.asarray(getattr(c, k)) - np.array(e[k])).max() < 1e-9, k


def test_arrangement_parser_errors_and_geometry():
    from upright_amd import arrangement as A

    ee = {"shape": "cuboid", "side_lengths": [0.3, 0.3, 0.02], "position": [0, 0, -0.01]}
    cube = {"mass": 1.0, "com_offset": [0, 0, 0], "shape": "cuboid", "side_lengths": [0.1, 0.1, 0.1]}

    def cfg(objects, contacts):
        return {"objects": {"ee": dict(ee), "cube": dict(cube)}, "arrangements": {"a": {"objects": objects, "contacts": contacts}}, "balancing": {"arrangement": "a"}}

    one = [{"name": "c1", "type": "cube", "parent": "ee"}]
    bodies, contacts = A.parse_control_objects(cfg(one, [{"first": "ee", "second": "c1", "mu": 0.5, "mu_margin": 0.1}]))
    assert np.allclose(bodies["c1"].com, [0, 0, 0.05]) and len(contacts) == 4 and abs(contacts[0].mu - 0.4) < 1e-15
    assert np.allclose(contacts[0].normal, [0, 0, -1])               # points into the first object (the tray)
    assert {tuple(np.round(c.r_co_o2, 9)) for c in contacts} == {(0.05, 0.05, 0), (0.05, -0.05, 0), (-0.05, 0.05, 0), (-0.05, -0.05, 0)}
    with pytest.raises(ValueError, match="too large"):             # math.py:149-158
        A.parse_control_objects(cfg(one, [{"first": "ee", "second": "c1", "mu": 0.5, "support_area_inset": 0.2}]))
    with pytest.raises(ValueError, match="Multiple control objects"):
        A.parse_control_objects(cfg(one + one, []))
    two = one + [{"name": "c2", "type": "cube", "parent": "ee", "offset": {"x": 0.2}}]
    with pytest.raises(ValueError, match="No contact points"):     # parsing.py:169: objects 0.1 m apart
        A.parse_control_objects(cfg(two, [{"first": "c1", "second": "c2", "mu": 0.5}]))
    with pytest.raises(ValueError, match="both"):
        A.support_offset({"r": 0.1})
    assert np.allclose(A.support_offset({"x": 0.1, "r": 0.2, "θ": "0.5pi"}), [0.1, 0.2])
    # stacking height of a tilted box: lowest corner touches the parent
    tilted = A.solid_of(cube, rotation=A.quat_xyzs_to_rot([0, np.sin(0.2), 0, np.cos(0.2)]))
    assert abs(tilted.exit_distance(np.array([0, 0, -1.0])) - 0.05 / np.cos(0.4)) < 1e-12
    # overlap polygon of two offset squares
    sq = np.array([[0, 0], [1, 0], [1, 1], [0, 1.0]])
    ov = A.overlap_polygon(sq, sq + 0.5)
    assert ov.shape == (4, 2) and np.allclose(sorted(map(tuple, ov)), [(0.5, 0.5), (0.5, 1), (1, 0.5), (1, 1)])
    assert A.overlap_polygon(sq, sq + 2.0) is None


def test_dynamic_obstacle_and_projectile_settings(arrangements):
    """BASELINE config 5 (ral23/experiments/projectile/_base.yaml:26-100 + obstacles/dynamic.yaml:1-36): one dynamic
    obstacle appends 9 entries to the state, named pairs may involve it and the ground, the projectile-path rows
    check the listed collision links."""
    import copy

    g = copy.deepcopy(json.load(open(GOLD / "configs.json"))["full_bottle_point1"]["controller"])
    g["obstacles"] = {
        "enabled": True, "minimum_distance": 0.1,
        "dynamic": [{"name": "projectile1", "radius": 0.2,
                     "modes": [{"time": 0, "position": [0, -10, 0], "velocity": [0, 0, 0], "acceleration": [0, 0, -9.81]}]}],
        "collision_pairs": [["wrist1_collision_link_0", "shoulder_collision_link_0"], ["wrist3_collision_link_0", "ground"],
                            ["balanced_object_collision_link_0", "projectile1"]],
    }
    g["projectile_path_constraint"] = {"enabled": True, "distances": [0.35], "scale": 0.2, "collision_links": ["balanced_object_collision_link"]}
    bodies, contacts = control.objects_from_fixture(arrangements["pink_bottle"])
    s = control.ControllerSettings(g, bodies=bodies, contacts=contacts)
    assert s.dims.o == 1 and s.dims.x() == 36 and s.initial_state.shape == (36,)
    assert np.array_equal(s.initial_state[27:], [0, -10, 0, 0, 0, 0, 0, 0, 0])       # static until observed (wrappers.py:378-383)
    P = control_bindings.problem_from_settings(s)
    assert (P.nx, P.nx_full, P.n_dyn) == (27, 36, 1)
    assert list(P.sph_frame) == [6, 4, 8, 9, -2] and list(P.pair_b) == [1, -1, 4] and list(P.pair_a) == [0, 2, 3]
    assert list(P.proj_sph) == [3] and np.allclose(P.proj_dist, [0.35]) and P.proj_scale == 0.2 and P.sph_r[4] == 0.2
    g["obstacles"]["collision_pairs"].append(["ground", "wrist1_collision_link_0"])
    s = control.ControllerSettings(g, bodies=bodies, contacts=contacts)
    with pytest.raises(RuntimeError, match="second object"):
        control_bindings.problem_from_settings(s)
    g["obstacles"]["collision_pairs"].pop()
    # a second dynamic obstacle (dimensions.h:32-45: 9 more state entries; system_pinocchio_mapping.h:84-97 loops over dims.o):
    # its sphere rides on the second 9-block, the projectile rows follow the LAST obstacle (projectile_path_constraint.h:82)
    g["obstacles"]["dynamic"].append({"name": "chair1", "radius": 0.25, "modes": [{"time": 0, "position": [1.5, 1, 0.25], "velocity": [0, 0, 0], "acceleration": [0, 0, 0]}]})
    g["obstacles"]["collision_pairs"].append(["base_collision_link_0", "chair1"])
    s = control.ControllerSettings(g, bodies=bodies, contacts=contacts)
    assert s.dims.o == 2 and s.dims.x() == 45 and np.array_equal(s.initial_state[36:39], [1.5, 1, 0.25])
    P2 = control_bindings.problem_from_settings(s)
    assert (P2.nx_full, P2.n_dyn) == (45, 2) and sorted(f for f in P2.sph_frame if f <= -2) == [-3, -2]
    assert P2.sph_frame[list(P2.sphere_names).index("chair1")] == -3 and P2.sph_frame[list(P2.sphere_names).index("projectile1")] == -2
    for k in range(control_bindings.MAX_DYNAMIC_OBSTACLES - 1):
        g["obstacles"]["dynamic"].append(dict(g["obstacles"]["dynamic"][1], name="chair%d" % (k + 2)))
    s = control.ControllerSettings(g, bodies=bodies, contacts=contacts)
    with pytest.raises(RuntimeError, match="dynamic obstacles"):
        control_bindings.problem_from_settings(s)


def test_target_trajectories_and_dimensions():
    d = control_bindings.OptimizationDimensions()
    d.robot.q = d.robot.v = 9; d.robot.x = 27; d.robot.u = 9; d.c = 4; d.nf = 3; d.o = 1
    assert (d.q(), d.v(), d.x(), d.f(), d.u()) == (12, 12, 36, 12, 21)
    r, Q = np.array([1.0, 2, 3]), np.array([0, 0, np.sin(0.3), np.cos(0.3)])
    cfg = {"waypoints": [{"time": 0, "position": [-2.0, 1.0, 0], "orientation": [0, 0, 0, 1]}, {"time": 2, "position": [0, 0, 1], "orientation": [0, 0, 0, 1]}]}
    t = control.TargetTrajectories.from_config(cfg, r, Q, np.zeros(3))
    assert np.allclose(t.xs[0], np.concatenate([r + [-2, 1, 0], Q, [0]])) and list(t.ts) == [0.0, 2.0]
    assert np.allclose(t.get_desired_state(1.0)[:3], r + [-1, 0.5, 0.5])   # midway between the waypoints
    assert np.allclose(t.get_desired_state(5.0)[:3], r + [0, 0, 1])
    a = control_bindings.vector_array(); a.push_back([1, 2]); assert a[0].dtype == np.float64


def test_core_bindings_data_model():
    b = core_bindings.RigidBody(2.0, np.diag([1.0, 2, 3]), [0.1, 0.2, 0.3])
    pvec = b.get_parameters()
    assert np.allclose(pvec, [2, 0.2, 0.4, 0.6, 1, 0, 0, 2, 0, 3])          # rigid_body.h:47-51
    b2 = core_bindings.RigidBody.from_parameters(pvec)
    assert np.allclose(b2.com, b.com) and np.allclose(b2.inertia, b.inertia)
    s = core_bindings.RigidBodyState.Zero()
    assert np.array_equal(s.pose.orientation, np.eye(3))
    c = core_bindings.ContactPoint(); c.object1_name, c.object2_name = "ee", "missing"
    with pytest.raises(IndexError):
        core_bindings.contact_tables({"box": b}, [c])


def test_chains():
    th = robots.thing()
    assert th.nq == 9 and [j.kind for j in th.joints[:3]] == [robots.PRISMATIC, robots.PRISMATIC, robots.REVOLUTE]
    ur = robots.ur10((-1.0, 1.0, 0.0))
    q = np.array([-1.0, 1.0, 0.0, 0.5 * np.pi, -0.25 * np.pi, 0.5 * np.pi, -0.25 * np.pi, 0.5 * np.pi, 0.417 * np.pi])
    p9, C9 = th.forward(q)
    p6, C6 = ur.forward(q[3:])
    assert np.allclose(p9, p6) and np.allclose(C9, C6)    # fixed base locked at base_pose (util.h:35-47)
    assert C9[2, 2] > 0.9998                               # tray level at home
    with pytest.raises(ValueError):
        robots.from_config({"base_type": "nonholonomic", "dims": {"q": 9}})


def test_c_abi_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "upright_mi.h").read_text()
    declared = set(re.findall(r"\b(upr_[a-z0-9_]+)\s*\(", header))
    declared -= {"upr_batch", "upr_problem"}
    lib = _capi.lib()
    bound = {name for name, _, _ in _capi.PROTOTYPES}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert getattr(lib, name) is not None
    assert C.sizeof(_capi.UprProblem) > 0
    if not lib.upr_device_available():
        # no GPU in this container: compute entry points must fail loudly, not fall back
        P = _capi.UprProblem(); P.nb = 1; P.nc = 1; P.nf = 3
        z = np.zeros(16)
        assert lib.upr_core_friction_rows(C.byref(P), 1, _capi.ptr(z), _capi.ptr(z)) != 0
        assert b"no HIP device" in lib.upr_last_error()


def test_struct_layout_matches_header():
    """ctypes mirror and C struct agree on the total size (field-order drift shows up here)."""
    src = '#include "include/upright_mi.h"\n#include <stdio.h>\nint main(){printf("%zu", sizeof(upr_problem));return 0;}\n'
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as td:
        f = Path(td) / "s.c"; f.write_text(src)
        subprocess.check_call(["gcc", "-I", str(ROOT), "-o", str(Path(td) / "s"), str(f)], cwd=ROOT)
        size = int(subprocess.check_output([str(Path(td) / "s")]))
    assert size == C.sizeof(_capi.UprProblem)


def test_shard_range_partitions():
    for total, world in ((1024, 8), (8192, 8), (10, 3), (5, 8), (0, 2)):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def _build_hpp_demo(tmp_path):
    import subprocess

    root = Path(__file__).resolve().parents[1]
    exe = tmp_path / "hpp_demo"
    lib = root / "upright_amd"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", f"-I{root / 'include'}", str(root / "tests" / "cpp" / "hpp_demo.cpp"),
                    f"-L{lib}", "-lupright_mi", f"-Wl,-rpath,{lib}", "-o", str(exe)], check=True)
    return exe


def _write_cpp_inputs(tmp_path, P, B, body_params, way, x0):
    import ctypes as C

    from upright_amd import _capi

    c = _capi.problem_to_c(P)
    (tmp_path / "problem.bin").write_bytes(bytes(memoryview((C.c_char * C.sizeof(c)).from_buffer(c))))
    (tmp_path / "body_params.bin").write_bytes(np.ascontiguousarray(body_params, dtype=np.float64).tobytes())
    (tmp_path / "way_p.bin").write_bytes(np.ascontiguousarray(way, dtype=np.float64).tobytes())
    (tmp_path / "x0.bin").write_bytes(np.ascontiguousarray(x0, dtype=np.float64).tobytes())


def test_cpp_header_twin_compiles_and_fails_loudly_without_gpu(arrangements, tmp_path):
    """include/upright_mi.hpp (C++ face of the C-ABI for ROS-node style callers) builds with -Wall -Werror against
    the in-tree library; on a machine without a GPU the constructor throws the library's error (no CPU path)."""
    import subprocess

    import torch

    from upright_amd.problem import thing_problem
    from upright_amd.sampling import level_tray_states, waypoints_for

    exe = _build_hpp_demo(tmp_path)
    P = thing_problem(arrangements["pink_bottle"], use_feedback_policy=True)
    x0 = level_tray_states(2, seed=3)
    _write_cpp_inputs(tmp_path, P, 2, np.broadcast_to(P.body_params, (2, 1, 10)), waypoints_for(P, x0), x0)
    r = subprocess.run([str(exe), str(tmp_path), "2"], capture_output=True, text=True)
    if not torch.cuda.is_available():
        assert r.returncode == 1 and "runtime_error" in r.stdout and "no HIP device" in r.stdout
    else:
        assert r.returncode == 0, r.stdout + r.stderr


def test_system_pinocchio_mapping_obstacle_coordinates_first():
    """Row a12.  bindings.SystemPinocchioMapping (pybindings.cpp:57-65) against hand-built vectors: the generalised
    position / velocity / acceleration of the Pinocchio model carry the dynamic obstacles' coordinates FIRST
    (dynamics/system_pinocchio_mapping.h:84-97, 104-116, 125-137), the robot's behind them, and the state is laid out
    [robot q, v, a | obstacle 0: r, v, a | obstacle 1: ...] (dimensions.h:32-45).  Then the call pattern of the
    reference's robot model (upright_control/robot.py:232-234: position from x alone, velocity and acceleration from (x, u))."""
    from upright_amd import control_bindings as cb

    for n_obs in (0, 1, 2):
        dims = cb.OptimizationDimensions()
        dims.robot.q, dims.robot.v, dims.robot.x, dims.robot.u = 9, 9, 27, 9
        dims.o, dims.c, dims.b, dims.nf = n_obs, 4, 1, 3
        assert dims.q() == 9 + 3 * n_obs and dims.v() == 9 + 3 * n_obs and dims.x() == 27 + 9 * n_obs
        m = cb.SystemPinocchioMapping(dims)
        q = 1.0 + np.arange(9.0); v = 10.0 + np.arange(9.0); a = 20.0 + np.arange(9.0)
        obs = [100.0 * (i + 1) + np.arange(9.0) for i in range(n_obs)]   # [r(3), v(3), a(3)] of obstacle i
        x = np.concatenate([q, v, a] + obs)
        u = np.concatenate([np.full(9, 7.0), np.zeros(dims.f())])        # jerk, forces: neither enters the mapping
        qp = m.get_pinocchio_joint_position(x)
        vp = m.get_pinocchio_joint_velocity(x, u)
        ap = m.get_pinocchio_joint_acceleration(x, u)
        assert qp.shape == (dims.q(),) and vp.shape == (dims.v(),) and ap.shape == (dims.v(),)
        for i in range(n_obs):
            assert np.array_equal(qp[3 * i: 3 * i + 3], obs[i][0:3])
            assert np.array_equal(vp[3 * i: 3 * i + 3], obs[i][3:6])
            assert np.array_equal(ap[3 * i: 3 * i + 3], obs[i][6:9])
        assert np.array_equal(qp[3 * n_obs:], q) and np.array_equal(vp[3 * n_obs:], v) and np.array_equal(ap[3 * n_obs:], a)
        # robot.py:225-234 -- forward_xu(x, u=None) substitutes a zero input of the model's velocity dimension
        u0 = np.zeros(dims.v())
        assert np.array_equal(m.get_pinocchio_joint_velocity(x, u0), vp)
        assert np.array_equal(m.get_pinocchio_joint_acceleration(x, u0), ap)
        # a state of the wrong length is an error (Eigen would read out of bounds: the mirror refuses)
        with pytest.raises(ValueError):
            m.get_pinocchio_joint_position(x[:-1])


def test_home_pose_and_the_arm_mount():
    """What the chain model of upright_amd/robots.py does to the reference's own scene: at the stock home
    configuration (thing.yaml:16) the signed distance d = |c_a - c_b| - r_a - r_b - minimum_distance of every collision pair of
    obstacles/simple.yaml:11-41, for the four right-angle yaws of the arm mount.  The shipped one (-pi/2) clears all 20 pairs
    and keeps home EE and the _point1 target within the arm's reach (_point1.yaml:1-2: "doable with either base or arm");
    the mount of rounds 1 - 3 (yaw 0) starts 0.16 m inside obstacle 3's margin -- a recorded number now, not a comment."""
    import json
    from pathlib import Path

    from upright_amd import robots
    from upright_amd.problem import THING_HOME

    cfg = json.load(open(Path(__file__).resolve().parent / "golden" / "configs.json"))["static_arch_point3"]["controller"]
    dmin = float(cfg["obstacles"]["minimum_distance"])
    assert len(cfg["obstacles"]["collision_pairs"]) == 20 == len(robots.SIMPLE_COLLISION_PAIRS)

    def scene(yaw):
        saved = robots.ARM_MOUNT_RPY.copy()
        try:
            robots.ARM_MOUNT_RPY[:] = (0.0, 0.0, yaw)
            ch = robots.thing()
        finally:
            robots.ARM_MOUNT_RPY[:] = saved
        cm = robots.collision_model(ch, robots.SIMPLE_COLLISION_PAIRS)
        R, o, fr = np.eye(3), np.zeros(3), {-1: (np.eye(3), np.zeros(3))}
        for i, (j, qi) in enumerate(zip(ch.joints, THING_HOME)):
            o = o + R @ j.p
            R = R @ j.R
            if j.kind == robots.REVOLUTE:
                R = R @ robots._axis_rot(j.axis, qi)
            else:
                o = o + R @ j.axis * qi
            fr[i] = (R.copy(), o.copy())
        fr[ch.nq] = (R @ ch.tool_R, o + R @ ch.tool_p)
        c = np.array([fr[f][1] + fr[f][0] @ off for f, off in zip(cm["sph_frame"], cm["sph_off"])])
        d = np.array([np.linalg.norm(c[a] - c[b]) - cm["sph_r"][a] - cm["sph_r"][b] - dmin for a, b in zip(cm["pair_a"], cm["pair_b"])])
        ee = ch.forward(THING_HOME)[0]
        shoulder = fr[2][1] + fr[2][0] @ ch.joints[3].p       # origin of shoulder_pan on the base at home
        reach = [np.linalg.norm((ee - shoulder)[:2]), np.linalg.norm((ee + np.array([-2.0, 1.0, 0.0]) - shoulder)[:2])]
        return d, reach

    d, reach = scene(float(robots.ARM_MOUNT_RPY[2]))
    assert robots.ARM_MOUNT_RPY[2] == -np.pi / 2
    assert d.min() > 0.29, d.min()                            # every pair clears minimum_distance at home, by 0.30 m or more
    assert max(reach) < 1.40, reach                           # home EE and the _point1 target within reach of the shoulder axis
    table = {yaw: scene(yaw) for yaw in (0.0, np.pi / 2, np.pi)}
    assert table[0.0][0].min() < -0.16 and int((table[0.0][0] < 0).sum()) == 3      # rounds 1 - 3: tray and wrist inside obstacle 3's margin
    for yaw in (0.0, np.pi / 2, np.pi):
        assert max(table[yaw][1]) > 1.8                       # ... and the other mounts put the _point1 target out of the arm's reach

def test_production_kernels_keep_their_registers_out_of_scratch():
    """Round 4: the SOFT instantiations of the headline family kept their lane-owned rows in scratch (313 / 404 / 196 spilled registers,
    one exposed reload per use) and the dense-Schur kernel an 18 x 18 factor per lane (811): -22 % and -12 % per QP launch once found
    (DESIGN.md "Registers that lived in scratch").  The property is the compiler's to break again, so it is asserted on the
    resource-usage remarks of EVERY translation unit of the production kernel (parts 0 .. 6, i.e. every shipped
    instantiation; hipcc cross-compiles gfx950 without a GPU).  Bounds = what the binary has today plus a few registers; where the
    scratch instructions sit is profiles/r05_scratch_by_line.txt (tools/scratch_by_line.py): in the headline kernels a dozen
    loop-carried scalars and four values around the contact factor of prep B; in the dice / cups kernels the exit code that writes
    the feedback gains (once per launch, outside the interior-point loop)."""
    import sys
    from concurrent.futures import ThreadPoolExecutor
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
    from kernel_resources import usage

    # template arguments (nq, nb, nc, nf, N, NT, ROWS, SOFT, DENSE) -> (spilled VGPRs, scratch bytes per lane) allowed
    bounds = {
        0: {"9ELi1ELi4ELi3ELi20ELi256ELb0ELb0ELb0": (16, 64), "9ELi1ELi4ELi3ELi20ELi256ELb1ELb0ELb0": (24, 96)},              # headline, with state-polytopic rows (round 6: 10 / 16 spilled; round 5: 21 / 26)
        1: {"9ELi1ELi4ELi3ELi20ELi256ELb0ELb1ELb0": (24, 96), "9ELi1ELi4ELi3ELi20ELi256ELb1ELb1ELb0": (48, 192),              # SOFT (313 / 404 / 196 before round 4)
            "9ELi1ELi4ELi1ELi20ELi256ELb0ELb1ELb0": (8, 32)},
        2: {"9ELi8ELi32ELi1ELi20ELi256ELb0ELb1ELb0": (64, 0)},                                                                 # upright_robust (spills go to the other register file: no scratch)
        3: {"9ELi3ELi16ELi3ELi20ELi256ELb1ELb0ELb1": (32, 0)},                                                                  # box_arch (811 before round 4; as for upright_robust: one workgroup per CU, what the allocator moves goes to the other register file -- scratch must stay 0)
        4: {"6ELi1ELi4ELi1ELi20ELi256ELb0ELb1ELb0": (8, 32), "6ELi1ELi4ELi1ELi10ELi256ELb0ELb1ELb0": (0, 0), "6ELi1ELi4ELi3ELi20ELi256ELb0ELb0ELb0": (8, 32)},
        5: {"9ELi2ELi8ELi3ELi20ELi256ELb0ELb0ELb1": (256, 1024), "9ELi7ELi28ELi3ELi20ELi256ELb0ELb0ELb0": (16, 448)},           # dice / cups: exit code only (see the docstring)
        6: {"9ELi8ELi32ELi1ELi100ELi256ELb0ELb1ELb0": (8, 0)},                                                                 # upright_robust at N = 100 (KFAR): fully unrolled sweeps spilled 448 B of addresses (13.3 ms per launch); unrolled in groups: none (9.5 ms)
    }
    with ThreadPoolExecutor(3) as ex:
        res = dict(zip(bounds, ex.map(usage, [str(k) for k in bounds])))
    for part, table in bounds.items():
        kernels = {n: u for n, u in res[part].items() if "upr_qp3_kernel" in n}
        assert len(kernels) == len(table), (part, list(kernels))
        for key, (spill_max, scratch_max) in table.items():
            name = [n for n in kernels if ("cfgILi" + key) in n]
            assert len(name) == 1, (part, key, list(kernels))
            u = kernels[name[0]]
            assert u["spill"] <= spill_max and u["scratch"] <= scratch_max, (part, key, u)


def test_constraint_kernels_keep_the_occupancy_they_were_sized_for():
    """Round 6: the linearisation kernel runs the headline batch in ONE round of 768 workgroups because three of them share a CU
    (<= 53 760 B of LDS each, set by the launcher, and <= 168 registers a lane, set by the compiler), the line search four two-wave
    workgroups per CU (<= 256 registers, no scratch in the shape without collision rows).  The registers are the compiler's to
    take: asserted on the resource-usage remarks of upr_api.hip (hipcc cross-compiles gfx950 without a GPU)."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
    from kernel_resources import usage

    res = usage("api")
    lin = {n: u for n, u in res.items() if "upr_linearize2_kernel" in n}
    assert len(lin) == 2, list(lin)                                        # nq = 6, 9
    for n, u in lin.items():
        assert u["vgpr"] + u.get("agpr", 0) <= 168 and u["spill"] == 0 and u["scratch"] == 0, (n, u)
    ls = {n: u for n, u in res.items() if "upr_linesearch_kernel" in n}
    assert len(ls) == 16, list(ls)                                         # nq x {exact, exact + rows, one body, generic} x {staged, not}
    for n, u in ls.items():
        assert u["vgpr"] + u.get("agpr", 0) <= 256, (n, u)
        if "ELi12ELi1ELb1ELb0" in n:                                       # the headline's contact structure, no collision rows
            assert u["spill"] == 0 and u["scratch"] == 0, (n, u)


def test_value_function_recursion_against_a_dense_solve():
    """upright_amd/value_function.py::riccati_value_function (the host-side Riccati recursion behind ControllerInterface.valueFunction*)
    on a small synthetic problem against the brute-force answer: the Hessian of the minimised quadratic in x_0 computed by condensing
    the whole horizon into one dense system (dynamics eliminated, softened equality rows as the penalty Z, barrier weights lam / t on
    HARD boxes and friction rows, proximal terminal equality).  No GPU, no oracle.  (Softened inequality rows are factored with a
    different weight, which needs the slack's own barrier pair: ControllerInterface refuses the queries for such problems --
    test_value_function_queries_refused_with_softened_inequality_rows.)"""
    import types

    from upright_amd.value_function import _dynamics, record_layout, riccati_value_function

    rng = np.random.default_rng(3)
    nq, nb, nc, nf, N, h = 2, 1, 1, 3, 4, 0.1
    nx, nfc, ne = 3 * nq, nf * nc, 6 * nb
    nu = nq + nfc
    P = types.SimpleNamespace(nq=nq, nx=nx, nu=nu, N=N, dt=h, nb=nb, nf=nf, nc=nc, pair_a=[], proj_sph=[], terminal_constraint=True,
                              slacks={"equality": True, "lower_L2_penalty": 100.0}, Qdiag=rng.uniform(0.1, 1.0, nx), Rdiag=rng.uniform(0.1, 1.0, nu),
                              xd=rng.normal(size=nx))
    o = record_layout(P)
    stride = o["hess"] + max(o["nh"], 3 * nq)   # (the terminal record keeps the 3 x nq position Jacobian in the Hessian slot)
    lin = np.zeros((N + 1, stride))
    iu = np.triu_indices(nq)
    Hee, Cs = [], []
    for k in range(N):
        J = rng.normal(size=(3, nq)); Hk = J.T @ J
        lin[k, o["hess"]:o["hess"] + o["nh"]] = Hk[iu]
        Ck = rng.normal(size=(ne, nx)); lin[k, o["gx"]:o["gx"] + ne * nx] = Ck.ravel()
        Hee.append(Hk); Cs.append(Ck)
    Jp = rng.normal(size=(3, nq)); lin[N, o["hess"]:o["hess"] + 3 * nq] = Jp.ravel()
    E = rng.normal(size=(5 * nc, nfc)); Df = rng.normal(size=(ne, nfc))
    ni = 2 * nx + 2 * nu + 5 * nc
    lam = rng.uniform(0.0, 2.0, (N + 1, ni)); t = rng.uniform(0.05, 2.0, (N + 1, ni))
    sol = dict(dx=np.zeros((N + 1, nx)), du=np.zeros((N, nu)), pi=rng.normal(size=(N + 1, nx)), nu=rng.normal(size=(N, ne)), lam=lam, slack=t)
    xs, us = rng.normal(size=(N + 1, nx)), rng.normal(size=(N, nu))
    Pk, pk, X, U = riccati_value_function(P, xs, us, lin, sol, E, Df)
    # ---- brute force: z = [x_0; u_0 .. u_{N-1}], x_k = Phi_k x_0 + sum Gamma_kj u_j; total quadratic 1/2 z' Hz z; V_xx = Schur complement in x_0
    A, Bq = _dynamics(nq, h)
    Bf = np.hstack([Bq, np.zeros((nx, nfc))])
    nz = nx + N * nu
    T = [np.hstack([np.eye(nx), np.zeros((nx, N * nu))])]          # x_k = T[k] z
    for k in range(N):
        Sel = np.zeros((nu, nz)); Sel[:, nx + k * nu:nx + (k + 1) * nu] = np.eye(nu)
        T.append(A @ T[k] + Bf @ Sel)
    w = lam / t
    Hz = np.zeros((nz, nz))
    D = np.hstack([np.zeros((ne, nq)), Df])
    for k in range(N):
        Sel = np.zeros((nu, nz)); Sel[:, nx + k * nu:nx + (k + 1) * nu] = np.eye(nu)
        Hxx = h * np.diag(P.Qdiag); Hxx[:nq, :nq] += h * Hee[k]
        if k >= 1:
            Hxx += np.diag(w[k][:nx] + w[k][nx:2 * nx])
        Huu = h * np.diag(P.Rdiag) + np.diag(w[k][2 * nx:2 * nx + nu] + w[k][2 * nx + nu:2 * nx + 2 * nu])
        Huu[nq:, nq:] += E.T @ (w[k][2 * nx + 2 * nu:, None] * E)
        R = Cs[k] @ T[k] + D @ Sel                                  # softened equality rows: penalty Z / 2 |C x + D u|^2
        Hz += T[k].T @ Hxx @ T[k] + Sel.T @ Huu @ Sel + 100.0 * R.T @ R
    CN = np.zeros((3 + 2 * nq, nx)); CN[:3, :nq] = -Jp; CN[3:, nq:] = np.eye(2 * nq)
    Hz += T[N].T @ (np.diag(w[N][:nx] + w[N][nx:2 * nx]) + CN.T @ CN / 1e-6) @ T[N]
    Hxx0, Hxu, Huu0 = Hz[:nx, :nx], Hz[:nx, nx:], Hz[nx:, nx:]
    V = Hxx0 - Hxu @ np.linalg.solve(Huu0, Hxu.T)
    assert np.abs(Pk[0] - V).max() < 1e-7 * np.abs(V).max(), (np.abs(Pk[0] - V).max(), np.abs(V).max())
    assert np.abs(Pk[N] - (np.diag(w[N][:nx] + w[N][nx:2 * nx]) + CN.T @ CN / 1e-6)).max() == 0.0
    assert np.array_equal(pk[2], sol["pi"][2])


def test_value_function_queries_refused_with_softened_inequality_rows(arrangements):
    """with HPIPM slacks on inequality rows (thing_demo.yaml, upright_robust's _base.yaml) the rebuilt cost-to-go would use
    lam / t where the kernels factor w0 (Z + gam / tau) / (Z + w0 + gam / tau): the three solver-level queries raise instead of
    answering wrongly; likewise with dynamic obstacles (no multiplier export), and before any solve on the current target."""
    g = json.load(open(GOLD / "configs.json"))
    for name, msg in (("thing_demo", "HPIPM slacks"), ("projectile_head_on", "dynamic obstacles"), ("full_bottle_point1", "no MPC solve yet")):
        ci = control_bindings.ControllerInterface(control.ControllerSettings(g[name]["controller"]))
        x = np.array(ci.settings.initial_state)
        for call in (lambda: ci.valueFunction(0.0, x), lambda: ci.valueFunctionStateDerivative(0.0, x)):
            with pytest.raises(RuntimeError, match=msg):
                call()
        # the equality multipliers need no barrier weights: with slacks that query only waits for a solve
        with pytest.raises(RuntimeError, match="no MPC solve yet" if name != "projectile_head_on" else msg):
            ci.stateInputEqualityConstraintLagrangian(0.0, x, np.zeros(ci.problem.nu))


# ---- the reference's own caller against the shims (tests/golden/make_caller_fixtures.py) ---------------------------------------------
def _tree_diff(a, b, path="", tol=None):
    """Paths at which two JSON trees differ (numbers under a `tol` prefix: to that absolute tolerance)."""
    out = []
    if isinstance(a, dict) and isinstance(b, dict):
        for k in sorted(set(a) | set(b)):
            if k not in a or k not in b:
                out.append(path + "/" + k + (" (missing here)" if k not in a else " (missing in the reference's)"))
            else:
                out += _tree_diff(a[k], b[k], path + "/" + k, tol)
    elif isinstance(a, list) and isinstance(b, list) and len(a) == len(b):
        for i, (x, y) in enumerate(zip(a, b)):
            out += _tree_diff(x, y, path + "/%d" % i, tol)
    elif a != b:
        t = next((v for p, v in (tol or {}).items() if path.startswith(p)), None)
        if not (t is not None and isinstance(a, float) and isinstance(b, float) and abs(a - b) <= t):
            out.append(f"{path}: {a!r} != {b!r}")
    return out


_CALLER_CONFIGS = ["ur10_demo", "thing_demo", "full_bottle_point1", "full_arch_point3", "static_arch_point3", "projectile_head_on",
                   "robust_sim", "sudden_t1.0", "full_cups_point1", "full_dice_point1", "full_bottle_arm_only"]


@pytest.mark.parametrize("name", _CALLER_CONFIGS)
def test_settings_field_for_field_against_the_references_wrappers(name):
    """Row P / b2 (VERDICT r05 missing 4): `settings_from_reference_wrappers.json` is what the REFERENCE's
    `wrappers.ControllerSettings.__init__` (wrappers.py:81-399, imported unmodified from /root/reference in the build container with
    `upright_control.bindings := upright_amd.control_bindings`, `upright_core.bindings := upright_amd.core_bindings`) leaves in this
    build's settings struct; `upright_amd.control.ControllerSettings` must fill the same struct identically from the same merged
    dict -- every field, exactly, except: the arrangement parser's lengths (its own arithmetic: 1e-15) and the URDF path (the
    reference compiles xacro files; the engine selects its chain by `robot.base_type` / `dims.q`)."""
    import standin

    ref = json.load(open(GOLD / "settings_from_reference_wrappers.json"))[name]
    c = json.load(open(GOLD / "configs.json"))[name]["controller"]
    ours = standin.dump_settings(control.ControllerSettings(c))
    for k in standin.SETTINGS_PATH_FIELDS:
        ours.pop(k), ref.pop(k)
    d = _tree_diff(ours, ref, tol={"/balancing_settings/bodies": 1e-15, "/balancing_settings/contacts": 1e-15})
    assert not d, d[:10]
    assert len(ref) >= 30 and ref["dims_totals"]["x"] == len(ref["initial_state"])   # (the fixture is not empty)


@pytest.mark.parametrize("name", ["thing_demo", "projectile_head_on", "ur10_demo"])
def test_manager_call_sequence_equals_the_references(name, monkeypatch):
    """`manager_call_sequence.json` records what the REFERENCE's `manager.ControllerManager` (manager.py:105-209: `from_config`,
    `warmstart`, 50 x `step` at a 4 ms simulator period against the 10 ms controller period, `get_mpc_trajectory`, `plan`) calls on
    `bindings.ControllerInterface`, argument by argument, and what it hands back per tick; `upright_amd.control.ControllerManager`
    driven by the same loop against the same recording stand-in must make the same calls with the same numbers."""
    import standin

    g = json.load(open(GOLD / "manager_call_sequence.json"))[name]
    c = json.load(open(GOLD / "configs.json"))[name]["controller"]
    monkeypatch.setattr(control_bindings, "ControllerInterface", standin.RecordingControllerInterface)
    mgr = control.ControllerManager.from_config(c)
    rec = mgr.mpc
    mgr.warmstart()
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
    assert mgr.timestep == g["timestep"]
    assert [cl[0] for cl in rec.calls] == [cl[0] for cl in g["calls"]]                 # the sequence itself
    # the target handed to reset(): the reference forms it with core.math.quat_multiply on the stub robot's pose
    d = _tree_diff(rec.calls, g["calls"], tol={"/1": 1e-14})
    assert not d, d[:10]
    assert not _tree_diff(outs, g["step_outputs"])
    assert list(mgr.replanning_times) == g["replanning_times"] and mgr.last_planning_time == g["last_planning_time"]
    assert [list(np.shape(v)) for v in (ts, xs, us)] == g["trajectory_shapes"]
    for k, v in (("ts", plan.ts), ("xs", plan.xs), ("us", plan.us)):
        assert np.array_equal(np.asarray(v), np.asarray(g["plan"][k])), k
    assert sum(cl[0] == "advanceMpc" for cl in g["calls"]) > 10                           # (it did re-plan)


def test_controller_model_kinematics_for_the_callers_logging():
    """manager.py:14-97 (`ControllerModel.update`, `angle_between_acc_and_normal`, `ddC_we_norm`) on the chain: velocity and
    classical acceleration of the tool link against central differences of the pose along q(t) = q + v t + a t^2 / 2."""
    c = json.load(open(GOLD / "configs.json"))["thing_demo"]["controller"]
    m = control.ControllerModel.from_config(c)
    rng = np.random.default_rng(5)
    q, v, a = rng.normal(size=9), rng.normal(size=9), rng.normal(size=9)
    m.update(np.concatenate([q, v, a]))
    p0, C0 = m.robot.link_pose(rotation_matrix=True)
    ch, h = m.robot.chain, 1e-4
    (pm, Cm), (pp, Cp) = ch.forward(q - v * h + 0.5 * a * h * h), ch.forward(q + v * h + 0.5 * a * h * h)
    assert np.allclose((pp - pm) / (2 * h), m.robot.link_velocity()[0], atol=1e-6)
    assert np.allclose((pp - 2 * p0 + pm) / h ** 2, m.robot.link_classical_acceleration()[0], atol=1e-5)
    W = (Cp - Cm) / (2 * h) @ C0.T
    assert np.allclose([W[2, 1], W[0, 2], W[1, 0]], m.robot.link_velocity()[1], atol=1e-6)
    ddC = (Cp - 2 * C0 + Cm) / h ** 2
    assert abs(np.linalg.norm(ddC, ord=2) - m.ddC_we_norm()) < 1e-4 * max(1.0, m.ddC_we_norm())
    m.update(np.concatenate([q, 0 * v, 0 * a]))
    assert abs(m.angle_between_acc_and_normal() - np.arccos(C0[2, 2])) < 1e-12           # at rest: the tray's tilt against gravity
