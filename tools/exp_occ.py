"""What would a THIRD workgroup per CU buy?  The headline shape at a horizon whose working set fits three workgroups into the LDS of a CU
(N = 10: 52 KB), compiled for two and for three workgroups per CU (-DUPR_QP3_OCC1=3: 168 registers a lane instead of 256).
python tools/exp_occ.py [N] [B]"""
import os, sys, subprocess, json
sys.path.insert(0, '.')
import numpy as np
N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 1536
if "--child" in sys.argv:
    from upright_amd.engine import BatchMPC
    from upright_amd.problem import thing_problem
    from upright_amd.sampling import level_tray_states, waypoints_for
    arr = json.load(open('tests/golden/arrangements.json'))
    P = thing_problem(arr['pink_bottle'], N=N)
    x0 = level_tray_states(B, seed=0); way = waypoints_for(P, x0, offset=(-0.004 * N * N, 0.002 * N * N, 0.0))   # (a target the short horizon can reach, as tools/dbg_n5.py)
    mpc = BatchMPC(P, B, way_p=way); mpc.set_observation(0.0, x0); mpc.advance()
    mpc.enable_timing(True)
    for _ in range(20):
        mpc.reset_async(); mpc.advance_async()
    mpc.sync()
    kt, st = mpc.kernel_times(), mpc.stats()
    _, xs, us = mpc.solution()
    np.savez(sys.argv[-1], xs=xs, qp_ms=kt["qp_ms"], its=st["qp_iters_last"], status=st["qp_status_last"], kernel=kt["qp_kernel"])
    sys.exit(0)
ref = None
for rep in range(2):
    for flags in ("-DUPR_QP3_OCC1=2", "-DUPR_QP3_OCC1=3"):
        f = "/tmp/exp_occ.npz"
        e = dict(os.environ, UPR_QP3_JIT="2", UPR_JIT_FLAGS=flags)
        subprocess.check_call([sys.executable, __file__, str(N), str(B), "--child", f], env=e, stderr=subprocess.DEVNULL)
        r = dict(np.load(f))
        if ref is None: ref = r
        print("N %d B %d %-20s qp %.4f ms -> %.0f k QPs/s | iterations mean %.2f max %d | status %s | max |dx| vs first %.1e" % (
            N, B, flags, r["qp_ms"], B / r["qp_ms"], r["its"].mean(), r["its"].max(), np.bincount(r["status"].astype(int)), float(np.abs(r["xs"] - ref["xs"]).max())))
