"""A/B of the production QP kernel: the ahead-of-time instantiation in libupright_mi.so against the run-time instantiation of the
CURRENT upr_qp3.h (UPR_QP3_JIT=2, optional UPR_JIT_FLAGS): plans compared bit for bit, QP launch times side by side.
python tools/exp_ab.py [headline|config4|config5|config3] [B]"""
import os, sys, subprocess, json
sys.path.insert(0, '.')
import numpy as np
name = sys.argv[1] if len(sys.argv) > 1 else "headline"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
if len(sys.argv) > 3 and sys.argv[3] == "child":
    import bench
    w = {"headline": lambda: bench.headline_workload(B), "config4": lambda: bench.config4_workload(B), "config5": lambda: bench.config5_workload(B),
         "config3": lambda: bench.config3_workload(B), "config5s": lambda: bench.config5_workload(B, slacks=True)}[name]()
    mpc = bench.make_engine(w)
    if name.startswith("config5"): mpc.set_projectile_flag(1.0)
    mpc.advance()
    mpc.enable_timing(True)
    for _ in range(10):
        mpc.reset_async(); mpc.advance_async()
    mpc.sync()
    kt = mpc.kernel_times(); _, xs, us = mpc.solution(); st = mpc.stats()
    np.savez(sys.argv[4], xs=xs, us=us, its=st["qp_iters_last"], status=st["qp_status_last"], qp_ms=kt["qp_ms"], kernel=kt["qp_kernel"])
    sys.exit(0)
out = {}
for tag, env in (("aot", {"UPR_QP3_JIT": "0"}), ("jit", {"UPR_QP3_JIT": "2"})):
    f = "/tmp/exp_ab_%s.npz" % tag
    e = dict(os.environ); e.update(env)
    subprocess.check_call([sys.executable, __file__, name, str(B), "child", f], env=e, stderr=subprocess.DEVNULL)
    out[tag] = np.load(f)
a, j = out["aot"], out["jit"]
print(name, "B", B, "| aot", str(a["kernel"])[:70], "%.4f ms" % a["qp_ms"], "| jit %.4f ms" % j["qp_ms"], "(%+.1f %%)" % (100 * (j["qp_ms"] / a["qp_ms"] - 1)))
print("  bit-identical plans:", bool(np.array_equal(a["xs"], j["xs"]) and np.array_equal(a["us"], j["us"])), "| iteration counts equal:", bool(np.array_equal(a["its"], j["its"])),
      "| max |dx|", float(np.abs(a["xs"] - j["xs"]).max()), "| status aot/jit", np.bincount(a["status"].astype(int)), np.bincount(j["status"].astype(int)))
