"""Config-3 experiment run (UPR_LIB=libupright_mi_exp1.so built with tools/exp_build.sh -DUPR_EXP_CONFIG3): timing + check against the oracle."""
import sys
sys.path.insert(0, '.')
import numpy as np
import bench
from oracle.oracle import Oracle
from upright_amd.sampling import stationary_guess
e = bench.time_extra(bench.config3_workload(4096), 2, 1, warm=(9, 3))
print("cold qp ms", e["kernel_ms"]["qp"], "warm qp ms", e["warm"]["kernel_ms"]["qp"], "warm its", e["warm"]["qp_iters_mean"], "conv", e["warm"]["qp_converged_fraction"], e["roofline"]["kernel"])
# parity: 4 instances, 4 SQP iterations, against the oracle
w = bench.config3_workload(4)
P = w["P"]; P.sqp_iters = 4
mpc = bench.make_engine(w)
mpc.advance()
_, xs, us = mpc.solution()
xs0, us0 = stationary_guess(w["x0"], P.N, P.nu)
xo, uo, so, _ = Oracle(P).solve_batch(0.0, w["x0"], xs0, us0, way_p=w["way"], nthreads=4)
print("max |x - x_oracle|", np.abs(xs - xo).max(), "max |u - u_oracle|", np.abs(us - uo).max(), "status", mpc.stats()["qp_status_last"], [s.qp_status_last for s in so])
