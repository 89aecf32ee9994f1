"""QP kernel time of the headline batch at a FIXED number of interior-point iterations (qp_tol = 0): experiment builds whose results are
not meaningful (e.g. -DUPR_QP3_EXP_SHAREG: every instance on instance 0's far arrays -- what would the kernel take if that traffic hit
the L2?) are compared on equal work.  python tools/exp_fixed.py [iters] [B]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import bench
its = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
w = bench.headline_workload(B)
P = w["P"]; P.qp_tol = 0.0; P.qp_iter_max = its
mpc = bench.make_engine(w)
mpc.advance()
mpc.enable_timing(True)
for _ in range(10):
    mpc.reset_async(); mpc.advance_async()
mpc.sync()
kt = mpc.kernel_times()
print("fixed", its, "iterations, B", B, "qp_ms %.4f" % kt["qp_ms"], "per iteration and round of 512: %.1f us" % (1e3 * kt["qp_ms"] / its / max(1, B / 512)))
