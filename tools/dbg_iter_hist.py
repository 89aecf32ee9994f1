import sys
sys.path.insert(0,'.')
import numpy as np, bench
w=bench.headline_workload(1024); mpc=bench.make_engine(w); mpc.advance()
it=mpc.stats()["qp_iters_last"].astype(int)
print(np.bincount(it))
np.save("gpurun_out/headline_iters.npy", it)
