"""Iteration-count histogram of the QP launch of a workload (headline by default) and the dispatch imbalance it implies."""
import sys, json
sys.path.insert(0, '.')
import numpy as np
import bench
B = 1024
w = bench.headline_workload(B)
mpc = bench.make_engine(w)
mpc.enable_timing(True)
for _ in range(3):
    mpc.reset(); mpc.advance()
st = mpc.stats()
its = st["qp_iters_last"].astype(int)
print("histogram", np.bincount(its), "mean", its.mean())
print("kernel", mpc.kernel_times())
bad = [79, 158, 180, 235, 495, 504, 520, 560, 618, 634, 636, 708, 816, 817]
print("oracle's capped instances on the GPU:", its[bad], st["qp_status_last"][bad])
np.save("gpurun_out/its_gpu.npy", its)
