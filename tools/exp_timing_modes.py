"""Cost of the engine's timing events: wall time per cold solve of the headline batch with upr_batch_enable_timing 0 / 3 / 2 / 1
(no events; around every fourth QP launch; around every QP launch; around every kernel), alternating, one box."""
import sys, time
sys.path.insert(0, '.')
import bench
w = bench.headline_workload(1024)
mpc = bench.make_engine(w)
for _ in range(5):
    mpc.reset_async(); mpc.advance_async()
mpc.sync()
K = 40
for rep in range(3):
    for mode in (0, 3, 2, 1):
        mpc.enable_timing(mode); mpc.sync()
        t0 = time.perf_counter()
        for _ in range(K):
            mpc.reset_async(); mpc.advance_async()
        mpc.sync()
        ms = 1e3 * (time.perf_counter() - t0) / K
        mpc.kernel_times()
        print("mode %d: %.4f ms per solve of the batch = %.0f solves/s" % (mode, ms, 1024 / ms * 1e3))
