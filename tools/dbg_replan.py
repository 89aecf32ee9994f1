"""The re-plan inside test_reference_call_sequence_other_configs (m.step(0.012, x0)): status, iterations and residuals of its QP for every
parametrisation, with the observed state (a) the start state again (what the test does) and (b) the plan's own state at t = 0.012."""
import sys, json, warnings
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
warnings.simplefilter("ignore")
import test_gpu_parity as T
arr = json.load(open('tests/golden/arrangements.json'))
cases = [("ur10_demo", {}, True),
         ("ur10_demo", {"mpc.time_horizon": 1.0, "waypoints": [{"time": 0, "position": [0.15, 0.1, 0.05], "orientation": [0, 0, 0, 1]}]}, True),
         ("thing_demo", {"sqp.hpipm.slacks.enabled": False}, True), ("thing_demo", {}, True), ("thing_demo", {}, False),
         ("thing_demo", {"sqp.hpipm.slacks.enabled": False}, False), ("full_bottle_point1", {}, False)]
for name, ov, level in cases:
    for mode in ("x0", "plan"):
        m = T._manager_from_golden(name, arr, **ov)
        P = m.mpc.problem
        x0 = np.array(m.settings.initial_state)
        if level:
            T._level_tool(P.chain, x0[:P.nq]); m.mpc._mpc.close(); m.mpc._mpc = None; m.mpc.reset(m.ref)
        m.warmstart()
        st0 = {k: v[0] for k, v in m.mpc._mpc.stats().items()}
        m.step(0.004, x0)
        xobs = x0 if mode == "x0" else np.array(m.step(0.008, x0)[0])
        m.step(0.012, xobs)
        st = {k: v[0] for k, v in m.mpc._mpc.stats().items()}
        print("%-20s level %-5s obs %-4s | first solve: status %d its %2d | re-plan: status %d its %2d res %.1e %.1e %.1e %.1e alpha %.3f viol %.2e" % (
            name, level, mode, st0["qp_status_last"], st0["qp_iters_last"], st["qp_status_last"], st["qp_iters_last"], st["qp_res_stat"], st["qp_res_eq"], st["qp_res_ineq"], st["qp_res_comp"], st["step_alpha_last"], st["constraint_violation"]))
