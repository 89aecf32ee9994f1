"""QP launch time of a bench workload against the interior-point iteration cap (1 .. 6): slope = one iteration of the whole batch,
intercept = set-up + exit (copy-in, first residuals, result / feedback gains out).  python tools/exp_itercost.py [headline|config5|config4] [B]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "headline"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
mk = {"headline": lambda: bench.headline_workload(B), "config4": lambda: bench.config4_workload(B), "config5": lambda: bench.config5_workload(B)}[name]
ts = []
for cap in (1, 2, 3, 4, 6):
    w = mk(); w["P"].qp_iter_max = cap
    mpc = bench.make_engine(w)
    if name.startswith("config5"): mpc.set_projectile_flag(1.0)
    mpc.advance(); mpc.enable_timing(True)
    for _ in range(10):
        mpc.reset_async(); mpc.advance_async()
    mpc.sync()
    ts.append((cap, mpc.kernel_times()["qp_ms"]))
    print(cap, "%.4f ms" % ts[-1][1])
a = np.polyfit([c for c, _ in ts], [t for _, t in ts], 1)
print("per iteration %.4f ms, intercept (set-up + exit) %.4f ms" % (a[0], a[1]))
if name == "headline":
    for fb in (True, False):
        w = mk(); w["P"].use_feedback_policy = fb
        mpc = bench.make_engine(w)
        mpc.advance(); mpc.enable_timing(True)
        for _ in range(10):
            mpc.reset_async(); mpc.advance_async()
        mpc.sync()
        print("use_feedback_policy", fb, "QP launch %.4f ms" % mpc.kernel_times()["qp_ms"])
