#!/usr/bin/env python3
"""Debug: BASELINE config 4 (bench.config4_workload) on the GPU against the oracle, SQP iteration by SQP iteration."""
import copy
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from upright_amd.engine import BatchMPC  # noqa: E402
from upright_amd.sampling import stationary_guess  # noqa: E402

print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a",
      "affinity:", len(os.sched_getaffinity(0)), "cpu_count:", os.cpu_count())
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
w = bench.config4_workload(B)
P, x0, way, bp = w["P"], w["x0"], w["way"], w["body_params"]
xs0, us0 = stationary_guess(x0, P.N, P.nu)
for it in (1, 2, 3):
    P.sqp_iters = it
    mpc = BatchMPC(P, B, way_p=way, body_params=bp)
    mpc.set_observation(0.0, x0)
    mpc.advance()
    st = mpc.stats()
    _, xs, us = mpc.solution()
    print("sqp", it, "gpu iters", st["qp_iters_last"][:4], "status", st["qp_status_last"][:4], "alpha", st["step_alpha_last"][:4],
          "res", [["%.1e" % st[k][b] for k in ("qp_res_stat", "qp_res_eq", "qp_res_ineq", "qp_res_comp")] for b in range(2)])
    for b in range(min(B, 2)):
        Pb = copy.copy(P); Pb.body_params = bp[b]; Pb.way_p = way[b]
        xo, uo, so, rc = Oracle(Pb).solve(0.0, x0[b], xs0[b], us0[b])
        print("   oracle", b, so.qp_iters_last, so.qp_status_last, so.step_alpha_last, "dx", np.abs(xs[b] - xo).max())
    mpc.close()
