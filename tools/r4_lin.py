"""Kernel times (linearise / QP / line search, HIP events) of one bench workload: python tools/r4_lin.py <config3|config4|config5|config5s|headline> [steps]
Environment switches are read by the library once per process (UPR_LIN_ROW_PASSES, UPR_QP_ORDER, UPR_LS_STAGE_FULL): one process per variant."""
import sys
sys.path.insert(0, '.')
import numpy as np
import bench
name = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w = {"config3": lambda: bench.config3_workload(4096), "config4": lambda: bench.config4_workload(1024), "config5": lambda: bench.config5_workload(1024),
     "config5s": lambda: bench.config5_workload(1024, slacks=True), "headline": lambda: bench.headline_workload(1024)}[name]()
mpc = bench.make_engine(w)
if name.startswith("config5"): mpc.set_projectile_flag(1.0)
mpc.advance()
mpc.enable_timing(True)
for _ in range(steps):
    mpc.reset_async(); mpc.advance_async()
mpc.sync()
kt = mpc.kernel_times()
print(name, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in kt.items()})
