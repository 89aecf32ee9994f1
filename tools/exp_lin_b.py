"""Linearisation / line-search launch time against the batch size (how many rounds of workgroups a launch takes).
python tools/exp_lin_b.py [headline|config5|config4] [B ...]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import bench
name = next((a for a in sys.argv[1:] if not a.isdigit()), "headline")
Bs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [128, 256, 384, 512, 585, 640, 768, 1024, 2048]
for B in Bs:
    w = {"headline": bench.headline_workload, "config5": bench.config5_workload, "config4": bench.config4_workload}[name](B)
    mpc = bench.make_engine(w)
    if name == "config5": mpc.set_projectile_flag(1.0)
    mpc.reset(); mpc.advance()
    mpc.enable_timing(True)
    for _ in range(10):
        mpc.reset_async(); mpc.advance_async()
    mpc.sync()
    kt = mpc.kernel_times()
    print("B %5d  workgroups(lin) %5d  linearize %.4f ms  qp %.4f  linesearch %.4f" % (B, -(-B * 21 // 24), kt["linearize_ms"], kt["qp_ms"], kt["linesearch_ms"]))
