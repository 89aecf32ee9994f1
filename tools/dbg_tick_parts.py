"""Where a closed-loop control period (configs[4], upr_batch_tick) goes: kernels (HIP events) against the whole period, with and
without the statistics copy, graph replay on / off (UPR_TICK_GRAPH)."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, bench
w = bench.config5_workload(1024)
for graph in ("1", "0"):
    os.environ["UPR_TICK_GRAPH"] = graph
    for stats in (True, False):
        mpc = bench.make_engine(w); mpc.set_projectile_flag(1.0)
        x, t = w["x0"].copy(), 0.0
        for _ in range(8):
            mpc.tick(t, x, want_stats=stats); t += 0.01
        t0 = time.perf_counter()
        for _ in range(200):
            mpc.tick(t, x, want_stats=stats); t += 0.01
        el = time.perf_counter() - t0
        mpc.enable_timing(True)
        for _ in range(20):
            mpc.tick(t, x, want_stats=stats); t += 0.01
        kt = mpc.kernel_times()
        ks = kt["linearize_ms"] + kt["qp_ms"] + kt["linesearch_ms"]
        print("graph", graph, "stats copy", stats, "| period %.4f ms | kernels (lin + qp + ls) %.4f = %.4f + %.4f + %.4f | rest %.4f ms | replays %s" % (
            1e3 * el / 200, ks, kt["linearize_ms"], kt["qp_ms"], kt["linesearch_ms"], 1e3 * el / 200 - ks, mpc.tick_graph_replays()))
        mpc.close()
