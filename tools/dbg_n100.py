"""configs[3] (upright_robust 8-corner) at the reference's own horizon, T = 10 s / N = 100 (upright_robust/config/demos/_base.yaml:62):
not instantiated in the production kernel, so the generic kernel runs it.  Kernel times of one SQP iteration of B instances."""
import sys
sys.path.insert(0, '.')
import numpy as np
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
w = bench.config4_workload(B)
w["P"].N = N
mpc = bench.make_engine(w)
mpc.enable_timing(True)
for _ in range(2):
    mpc.reset(); mpc.advance()
st = mpc.stats()
print("B", B, "N", N, mpc.kernel_times(), "qp iters mean", {k: float(v.mean()) for k, v in st.items()})
