#!/usr/bin/env python3
"""Wall-clock latency of one MPC solve (sqp_iteration = 1, cold start, host call to host return) for small batches of the
headline configuration: the drop-in case is B = 1 (one controller).  usage: python tools/latency_small_batches.py"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from upright_amd.engine import BatchMPC  # noqa: E402
from upright_amd.problem import thing_problem  # noqa: E402
from upright_amd.sampling import level_tray_states, waypoints_for  # noqa: E402

arr = json.load(open(ROOT / "tests" / "golden" / "arrangements.json"))
P = thing_problem(arr["pink_bottle"], use_feedback_policy=True)
for B in (1, 8, 64, 256, 512, 1024, 4096):
    x0 = level_tray_states(B, seed=0)
    mpc = BatchMPC(P, B, way_p=waypoints_for(P, x0))
    mpc.set_observation(0.0, x0); mpc.advance()
    ts = []
    for _ in range(20):
        mpc.reset()
        t = time.perf_counter()
        mpc.set_observation(0.0, x0); mpc.advance()
        ts.append(time.perf_counter() - t)
    ts = np.array(ts) * 1e3
    print(f"B={B:5d}  solve latency median {np.median(ts):7.3f} ms  min {ts.min():7.3f} ms  device {mpc.last_solve_ms():7.3f} ms  -> {B / np.median(ts) * 1e3:9.0f} solves/s")
    mpc.close()
