"""First tick of test_closed_loop_mpc's loop at which a QP does not converge, for several seeds (scenario design aid)."""
import sys, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from upright_amd.engine import BatchMPC
from upright_amd.problem import thing_problem
from upright_amd.sampling import level_tray_states, waypoints_for
arr = json.load(open('tests/golden/arrangements.json'))
for seed in (51, 52, 53, 54, 55, 56):
    B = 4
    P = thing_problem(arr["pink_bottle"])
    x0 = level_tray_states(B, seed=seed); way = waypoints_for(P, x0)
    mpc = BatchMPC(P, B, way_p=way)
    x = x0.copy(); t, dt = 0.0, 0.01
    bad = []
    for tick in range(60):
        mpc.set_observation(t, x); mpc.advance()
        st = mpc.stats()
        if np.any(st["qp_status_last"] != 0): bad.append((tick, st["qp_status_last"].astype(int).tolist(), ["%.1e" % v for v in st["qp_res_eq"]]))
        _, u = mpc.evaluate(t)
        j = u[:, :9]; q, v, a = x[:, :9], x[:, 9:18], x[:, 18:]
        x = np.concatenate([q + dt * v + dt ** 2 / 2 * a + dt ** 3 / 6 * j, v + dt * a + dt ** 2 / 2 * j, a + dt * j], axis=1); t += dt
    print("seed", seed, "failing ticks", bad[:4], len(bad))
    mpc.close()
