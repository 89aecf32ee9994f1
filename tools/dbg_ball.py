import sys, json, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_emu import _projectile_case
from upright_amd.engine import BatchMPC
arrs = json.load(open('tests/golden/arrangements.json'))
B = 4
P, x0r, way, _, _, dyn = _projectile_case(arrs, B, use_feedback_policy=True)
for flag in (1.0, 0.0):
    mpc = BatchMPC(P, B, way_p=way)
    mpc.set_projectile_flag(flag)
    x = np.concatenate([x0r, dyn], axis=1)
    t, dt = 0.0, 0.01
    dumped = flag != 1.0
    nfail = np.zeros(B, dtype=int)
    for tick in range(150):
        mpc.set_observation(t, x)
        mpc.advance()
        st = mpc.stats()
        _, u = mpc.evaluate(t, x_obs=x)
        _, xs, us = mpc.solution()
        bad = ~np.isfinite(u).all(axis=1)
        nfail += (st["qp_status_last"] != 0)
        if bad.any() or tick % 25 == 0 or (st["qp_status_last"] != 0).any():
            print(flag, tick, 'bad', bad, 'status', st["qp_status_last"], 'iters', st["qp_iters_last"], 'alpha', st["step_alpha_last"], 'xsfin', np.isfinite(xs).all(), np.isfinite(us).all(), 'res', st["qp_res_stat"], st["qp_res_eq"])
            K = mpc.feedback_gains()
            print('   K finite', np.isfinite(K).all(axis=(1,2,3)))
        if (st["qp_status_last"] != 0).any() and not dumped:
            np.savez('gpurun_out/ball_first_failure.npz', tick=tick, t=t, x=x, status=st["qp_status_last"], xs=xs, us=us, flag=flag, way=way, dyn=dyn)
            dumped = True
        if bad.any(): break
        j = u[:, :9]
        q, v, a = x[:, :9], x[:, 9:18], x[:, 18:27]
        ro, vo, ao = x[:, 27:30], x[:, 30:33], x[:, 33:36]
        x = np.concatenate([q + dt * v + dt ** 2 / 2 * a + dt ** 3 / 6 * j, v + dt * a + dt ** 2 / 2 * j, a + dt * j, ro + dt * vo + 0.5 * dt * dt * ao, vo + dt * ao, ao], axis=1)
        t += dt
    print('flag', flag, 'ticks with qp_status != 0 per instance:', nfail)
    mpc.close()
