"""Headline workload at other batch sizes: solves/s and ms per solve of the batch (device time, cold start, feedback policy on)."""
import sys
sys.path.insert(0, '.')
import bench
for B in (64, 256, 512, 1024, 2048, 4096, 8192):
    e = bench.time_extra(bench.headline_workload(B), 10, 3)
    print("B %5d  %9.0f solves/s  %7.3f ms per step  qp %7.3f ms  its %.2f" % (B, e["value"], e["ms_per_step"], e["kernel_ms"]["qp"], e["qp_iters_mean"]))
