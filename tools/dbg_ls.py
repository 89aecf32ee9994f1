"""Phase cycles of the line-search kernel (instrumented build: tools/build_prof.sh lin; UPR_LIB=libupright_mi_prof.so)."""
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np
import bench
from upright_amd import _capi
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = bench.headline_workload(B)
mpc = bench.make_engine(w)
lib = _capi.lib()
out = np.zeros(16)
mpc.reset(); mpc.advance()
lib.upr_debug_ls_prof(out.ctypes.data_as(C.POINTER(C.c_double)), 1)
mpc.enable_timing(True)
for _ in range(3):
    mpc.reset(); mpc.advance()
mpc.sync()
lib.upr_debug_ls_prof(out.ctypes.data_as(C.POINTER(C.c_double)), 0)
lab = ["record copy, ranking, barrier", "(entry)", "staging of xs, dx, us, du", "baseline knot terms", "two reductions", "trial: targets, Xt / Ut, sin / cos, barrier",
       "trial: knot terms (chain walk)", "trial: reduction, acceptance", "apply step, statistics", "(return)", "remembered solution"]
n = out[15]; prev = 0.0
for i, l in enumerate(lab):
    t = out[i] / n
    print("%-46s %8.0f cycles (at %8.0f)" % (l, t - prev, t)); prev = t
print("workgroups", n, mpc.kernel_times())
