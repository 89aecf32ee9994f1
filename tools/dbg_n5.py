import sys, os, json; sys.path.insert(0, ".")
import numpy as np
from upright_amd.engine import BatchMPC
from upright_amd.problem import thing_problem
from upright_amd.sampling import level_tray_states, stationary_guess, waypoints_for
from oracle.oracle import Oracle
arr = json.load(open("tests/golden/arrangements.json"))
B, N = 5, 5
P = thing_problem(arr["pink_bottle"], N=N, use_feedback_policy=True)
x0 = level_tray_states(B, seed=90 + N); way = waypoints_for(P, x0, offset=(-0.004 * N * N, 0.002 * N * N, 0.0))
xs0, us0 = stationary_guess(x0, N, P.nu)
xo, uo, so, _ = Oracle(P).solve_batch(0.0, x0, xs0, us0, way_p=way, nthreads=1)
print("oracle", [s.qp_status_last for s in so], [s.qp_iters_last for s in so], ["%.1e" % max(s.qp_res) for s in so])
for kern in ("", "1"):
    if kern: os.environ["UPR_QP_KERNEL"] = kern
    mpc = BatchMPC(P, B, way_p=way); mpc.set_observation(0.0, x0); mpc.advance(); st = mpc.stats()
    _, xs, us = mpc.solution()
    print(mpc.kernel_times()["qp_kernel"], st["qp_status_last"], st["qp_iters_last"], st["qp_res_stat"], "max|x-xo|", np.abs(xs - xo).max())
    mpc.close()
