"""The headline QP at 128 lanes a workgroup against 256 (run-time instantiations, UPR_JIT_NT): time of ONE round (B = 512: two workgroups per CU
either way) and of the bench batch.  What a 128-lane / three-per-CU kernel would have to beat.  python tools/exp_nt.py"""
import os, sys, subprocess, json
sys.path.insert(0, '.')
import numpy as np
if "--child" in sys.argv:
    import bench
    B = int(sys.argv[-2])
    w = bench.headline_workload(B)
    mpc = bench.make_engine(w)
    mpc.advance(); mpc.enable_timing(True)
    for _ in range(20):
        mpc.reset_async(); mpc.advance_async()
    mpc.sync()
    kt, st = mpc.kernel_times(), mpc.stats()
    _, xs, us = mpc.solution()
    np.savez(sys.argv[-1], xs=xs, qp_ms=kt["qp_ms"], its=st["qp_iters_last"], kernel=kt["qp_kernel"])
    sys.exit(0)
for B in (512, 1024):
    ref = None
    for nt in ("256", "128"):
        f = "/tmp/exp_nt.npz"
        e = dict(os.environ, UPR_QP3_JIT="2", UPR_JIT_NT=nt)
        subprocess.check_call([sys.executable, __file__, "--child", str(B), f], env=e, stderr=subprocess.DEVNULL)
        r = dict(np.load(f))
        if ref is None: ref = r
        print("B %4d NT %s: qp %.4f ms (%+.1f %%) | iterations mean %.2f | max |dx| vs NT 256 %.1e | %s" % (B, nt, r["qp_ms"], 100 * (r["qp_ms"] / ref["qp_ms"] - 1), r["its"].mean(), float(np.abs(r["xs"] - ref["xs"]).max()), str(r["kernel"])[:60]))
