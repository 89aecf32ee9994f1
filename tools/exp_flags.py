"""A/B/C... of run-time instantiations of the CURRENT upr_qp3.h under different compiler flags (UPR_QP3_JIT=2, UPR_JIT_FLAGS): QP launch
times side by side, plans and iteration counts compared with the first flag set's.
python tools/exp_flags.py <headline|config4|config5|config5s|config3|headline_r03> <B> "<flags 0>" "<flags 1>" ...   ("" = no flags)"""
import os, sys, subprocess
sys.path.insert(0, '.')
import numpy as np
name, B = sys.argv[1], int(sys.argv[2])
if sys.argv[3] == "--child":
    import bench
    w = {"headline": lambda: bench.headline_workload(B), "config4": lambda: bench.config4_workload(B), "config5": lambda: bench.config5_workload(B),
         "config3": lambda: bench.config3_workload(B), "config5s": lambda: bench.config5_workload(B, slacks=True),
         "headline_r03": lambda: bench.headline_r03_geometry_workload(B)}[name]()
    mpc = bench.make_engine(w)
    if name.startswith("config5"): mpc.set_projectile_flag(1.0)
    mpc.advance()
    mpc.enable_timing(True)
    reps = 20
    for _ in range(reps):
        mpc.reset_async(); mpc.advance_async()
    mpc.sync()
    kt = mpc.kernel_times(); _, xs, us = mpc.solution(); st = mpc.stats()
    np.savez(sys.argv[4], xs=xs, us=us, its=st["qp_iters_last"], status=st["qp_status_last"], qp_ms=kt["qp_ms"], kernel=kt["qp_kernel"])
    sys.exit(0)
ref = None
for i, flags in enumerate(sys.argv[3:]):
    f = "/tmp/exp_flags_%d.npz" % i
    e = dict(os.environ, UPR_QP3_JIT="2", UPR_JIT_FLAGS=flags)
    subprocess.check_call([sys.executable, __file__, name, str(B), "--child", f], env=e, stderr=subprocess.DEVNULL)
    r = np.load(f)
    if ref is None: ref = r
    print("%-14s %-60s qp %.4f ms (%+.1f %%) | bit-identical to [0]: %s | iteration counts equal: %s (mean %.3f max %d) | max |dx| %.2e | status %s" % (
        name, repr(flags), r["qp_ms"], 100 * (r["qp_ms"] / ref["qp_ms"] - 1), bool(np.array_equal(ref["xs"], r["xs"]) and np.array_equal(ref["us"], r["us"])),
        bool(np.array_equal(ref["its"], r["its"])), r["its"].mean(), r["its"].max(), float(np.abs(ref["xs"] - r["xs"]).max()), np.bincount(r["status"].astype(int))))
