import sys; sys.path.insert(0,'.')
import bench
from upright_amd.problem import thing_problem
for fb in (True, False):
    w = bench.headline_workload(1024)
    w["P"].use_feedback_policy = fb
    e = bench.time_extra(w, 10, 2)
    print("use_feedback_policy", fb, "qp ms", e["kernel_ms"]["qp"], "step ms", e["ms_per_step"])
