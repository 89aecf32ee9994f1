"""Phase cycles of the linearisation kernel (instrumented build: hipcc ... -DUPR_LIN_PROF -o upright_amd/libupright_mi_prof.so;
run with UPR_LIB=libupright_mi_prof.so)."""
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np
import bench
from upright_amd import _capi
name = sys.argv[1] if len(sys.argv) > 1 else "headline"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
w = {"config3": lambda: bench.config3_workload(B if len(sys.argv) > 2 else 4096), "config4": lambda: bench.config4_workload(B), "config5": lambda: bench.config5_workload(B),
     "headline": lambda: bench.headline_workload(B)}[name]()
mpc = bench.make_engine(w)
if name == "config5": mpc.set_projectile_flag(1.0)
lib = _capi.lib()
out = np.zeros(8)
mpc.reset(); mpc.advance()
lib.upr_debug_lin_prof(out.ctypes.data_as(C.POINTER(C.c_double)), 1)
mpc.enable_timing(True)
for _ in range(3):
    mpc.reset(); mpc.advance()
mpc.sync()
lib.upr_debug_lin_prof(out.ctypes.data_as(C.POINTER(C.c_double)), 0)
names = ["phase 0: stage x, u, sin/cos, Df f (lin2: record prefix, sin/cos)", "phase 1a: the value walk (lin2: beside Df f, targets)", "phase 1: tangents from the snapshots, residual, stores", "collision rows + barrier (lin2: values, position errors)", "phase 2: MFMA Hessian, gradient", "collision rows a: sphere centres"]
n = out[7]
for i, l in enumerate(names):
    print("%-42s %9.0f cycles / workgroup" % (l, out[i] / n))
print("workgroups", n, mpc.kernel_times())
