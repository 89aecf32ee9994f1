"""Phase cycles of the line-search kernel (instrumented build: tools/build_prof.sh with -DUPR_LS_PROF; UPR_LIB=libupright_mi_prof.so)."""
import sys
sys.path.insert(0, '.')
import numpy as np
import bench
w = bench.headline_workload(1024)
mpc = bench.make_engine(w)
mpc.enable_timing(True)
for _ in range(3):
    mpc.reset(); mpc.advance()
st = mpc.stats()
names = ["qp_res_stat", "qp_res_eq", "qp_res_ineq", "qp_res_comp"]
lab = ["baseline terms + reduction", "descent / norms + reduction", "trial evaluation(s) + reduction", "apply step, statistics"]
for n, l in zip(names, lab):
    print("%-36s mean %9.0f  max %9.0f cycles" % (l, st[n].mean(), st[n].max()))
print(mpc.kernel_times())
