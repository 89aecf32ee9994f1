"""Configs 3 and 4 on an experiment build (tools/exp_build.sh -DUPR_EXP_CONFIG3; UPR_LIB=libupright_mi_exp1.so)."""
import sys
sys.path.insert(0, '.')
import bench
e = bench.time_extra(bench.config3_workload(4096), 2, 1, warm=(9, 3))
print("config 3: cold qp ms %.2f warm qp ms %.2f conv %.2f" % (e["kernel_ms"]["qp"], e["warm"]["kernel_ms"]["qp"], e["warm"]["qp_converged_fraction"]))
e = bench.time_extra(bench.config4_workload(1024), 3, 1)
print("config 4: qp ms %.3f conv %.2f its %.2f" % (e["kernel_ms"]["qp"], e["qp_converged_fraction"], e["qp_iters_mean"]))
