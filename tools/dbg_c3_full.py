"""configs[2] at full size from the stock home pose: converged fraction and constraint violation against the SQP iteration budget."""
import sys
sys.path.insert(0, '.')
import numpy as np
import bench
from upright_amd.engine import BatchMPC
B = 4096
w = bench.config3_workload(B)
P = w["P"]
for iters in (12, 15, 20, 30):
    mpc = BatchMPC(P, B, way_p=w["way"])
    mpc.set_sqp_iterations(iters)
    mpc.set_observation(0.0, w["x0"])
    mpc.advance()
    st = mpc.stats()
    v = st["constraint_violation"]
    print(iters, "status", np.bincount(st["qp_status_last"].astype(int), minlength=3), "sqp done hist", np.bincount(st["sqp_iters_done"].astype(int))[-6:],
          "viol < 1e-4: %.4f" % np.mean(v < 1e-4), "viol quantiles", np.quantile(v, [0.5, 0.9, 0.99, 1.0]))
    mpc.close()
