"""Closed-loop tick (configs[4]) with and without the per-kernel event timing: what the instrumentation costs per tick."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, bench
w = bench.config5_workload(1024)
for timing in (False, True, False):
    mpc = bench.make_engine(w); mpc.set_projectile_flag(1.0)
    x, t = w["x0"].copy(), 0.0
    mpc.tick(t, x, want_stats=True)
    mpc.enable_timing(timing)
    t0 = time.perf_counter()
    for _ in range(100):
        _, u, st = mpc.tick(t, x, want_stats=True); t += 0.01
    el = time.perf_counter() - t0
    kt = mpc.kernel_times() if timing else None
    print("timing", timing, "ms per tick %.4f" % (10 * el), (kt["linearize_ms"] + kt["qp_ms"] + kt["linesearch_ms"]) if kt else "")
    mpc.close()
