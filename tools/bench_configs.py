#!/usr/bin/env python3
"""Timing of the other BASELINE configurations (parity-test cases of bench.py's headline workload): one MPC solve
(sqp_iteration = 1) of a batch from cold start, device time only.  usage: python tools/bench_configs.py [B]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from upright_amd import robots  # noqa: E402
from upright_amd.engine import BatchMPC  # noqa: E402
from upright_amd.problem import THING_HOME, thing_problem  # noqa: E402
from upright_amd.sampling import level_tray_states, waypoints_for  # noqa: E402

arr = json.load(open(ROOT / "tests" / "golden" / "arrangements.json"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024


def run(name, P, x0, way, body_params=None, steps=5):
    mpc = BatchMPC(P, len(x0), way_p=way, body_params=body_params)
    mpc.set_observation(0.0, x0)
    mpc.advance()
    mpc.sync()
    t = time.perf_counter()
    for _ in range(steps):
        mpc.reset_async(); mpc.advance_async()
    mpc.sync()
    ms = 1e3 * (time.perf_counter() - t) / steps
    st = mpc.stats()
    print(f"{name:44s} B={len(x0):5d} nx={P.nx} nu={P.nu} eq={6 * P.nb} pairs={len(P.pair_a):2d}  {ms:8.2f} ms/solve-batch  "
          f"{len(x0) / ms * 1e3:9.0f} solves/s  qp iters {st['qp_iters_last'].mean():.1f}  converged {np.mean(st['qp_status_last'] == 0):.2f}")
    mpc.close()


def level(chain, q):
    _, C = chain.forward(q)
    ez = np.array([0.0, 0.0, 1.0]); v = C.T @ ez; ax = np.cross(ez, v); s, c = np.linalg.norm(ax), ez @ v
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    chain.tool_R = chain.tool_R @ (np.eye(3) + K + K @ K * ((1 - c) / (s * s)))


rng = np.random.default_rng(0)
# config 2 (headline)
P = thing_problem(arr["pink_bottle"], use_feedback_policy=True)
x0 = level_tray_states(B, seed=0)
run("config 2: Thing + pink_bottle (headline)", P, x0, waypoints_for(P, x0))
# config 3: box_arch + 20 collision pairs, waypoint _point3
P = thing_problem(arr["box_arch"])
for k, v in robots.collision_model(P.chain, robots.SIMPLE_COLLISION_PAIRS).items():
    setattr(P, k, v)
x0 = np.tile(np.concatenate([THING_HOME, np.zeros(18)]), (B, 1)); x0[:, 1] = 0.3 + rng.uniform(-0.05, 0.05, B)
run("config 3: box_arch + static obstacles", P, x0, waypoints_for(P, x0, offset=(0.0, -2.0, 0.25)))
P = thing_problem(arr["box_arch"])
run("          box_arch without obstacles", P, x0, waypoints_for(P, x0, offset=(0.0, -2.0, 0.25)))
# config 4: robust 8-corner arrangement, per-instance inertial parameters, frictionless
P = thing_problem(arr["robust_8corner"], nf=1, force_weight=0.0)
level(P.chain, THING_HOME)
bp = np.zeros((B, 8, 10))
for b in range(B):
    sc = (1.0, 0.5, 0.1)[b % 3]
    for i in range(8):
        com = rng.uniform([-0.06, -0.06, -0.15], [0.06, 0.06, 0.15])
        bp[b, i] = [1.0, *com, sc * 0.009375, 0, 0, sc * 0.009375, 0, sc * 0.00375]
x0 = np.tile(np.concatenate([THING_HOME, np.zeros(18)]), (B, 1)); x0[:, :2] += rng.uniform(-0.25, 0.25, (B, 2))
run("config 4: robust 8-corner (per-instance params)", P, x0, waypoints_for(P, x0, offset=(-0.5, 0.5, 0.0)), body_params=bp)
# config 2': frictionless pink_bottle
P = thing_problem(arr["pink_bottle"], nf=1)
level(P.chain, THING_HOME)
run("config 2': Thing + pink_bottle frictionless", P, x0, waypoints_for(P, x0, offset=(-0.5, 0.5, 0.0)))
# config 5: Thing + bottle, thrown ball (dynamic obstacle) + self-collision / ground pairs + projectile-path row
sys.path.insert(0, str(ROOT / "tests"))
from test_emu import _projectile_case  # noqa: E402

P, x0r, way, _, _, dyn = _projectile_case(arr, B, use_feedback_policy=True)
mpc = BatchMPC(P, B, way_p=way)
mpc.set_projectile_flag(1.0)
x0f = np.concatenate([x0r, dyn], axis=1)
mpc.set_observation(0.0, x0f); mpc.advance(); mpc.sync()
t = time.perf_counter()
for _ in range(5):
    mpc.reset_async(); mpc.set_observation(0.0, x0f); mpc.advance_async()
mpc.sync()
ms = 1e3 * (time.perf_counter() - t) / 5
st = mpc.stats()
print(f"{'config 5: thrown ball (3 pairs + projectile row)':44s} B={B:5d} nx={P.nx}+9 nu={P.nu} eq=6 pairs={len(P.pair_a):2d}  {ms:8.2f} ms/solve-batch  "
      f"{B / ms * 1e3:9.0f} solves/s  qp iters {st['qp_iters_last'].mean():.1f}  converged {np.mean(st['qp_status_last'] == 0):.2f}")
mpc.close()
