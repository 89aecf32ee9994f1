"""CPU tests of the oracle: pinned against the golden fixtures generated from the reference's Python and
against the known answers the reference's own tests hold (upright_core/tests/test_parsing.py:81-182,
arrangements.yaml:184), plus a numpy twin, finite differences and physics known answers for the parts of
the path nothing in the reference pins."""
import json
from pathlib import Path

import numpy as np
import pytest
from scipy.optimize import linprog, minimize

from oracle.oracle import Oracle
from upright_amd.problem import THING_HOME, thing_problem
from upright_amd.sampling import level_tray_states, stationary_guess, waypoints_for

G = 9.81


def skew(v):
    x, y, z = v
    return np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])


# ---------------------------------------------------------------------------------------------------------
# golden fixtures vs the reference's own test expectations
def _unordered_close(a, b, tol=1e-9):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and all(np.min(np.linalg.norm(b - r, axis=1)) < tol for r in a)


def test_fixture_box_known_answers(arrangements):
    a = arrangements["tests/box"]  # test_parsing.py:81-111
    (body,) = a["bodies"]
    assert np.isclose(body["mass"], 1.0) and np.allclose(body["com"], [0, 0, 0.1])
    assert np.allclose(body["inertia"], np.diag([0.2 ** 2 + 0.2 ** 2] * 3) / 12.0)
    assert len(a["contacts"]) == 4
    for c in a["contacts"]:
        assert np.allclose(c["normal"], [0, 0, -1]) and np.allclose(np.array(c["span"]) @ c["normal"], 0)
        assert np.isclose(c["mu"], 0.45) and c["object1_name"] == "ee" and c["object2_name"] == "box"
    exp = [[0.1, 0.1, 0], [0.1, -0.1, 0], [-0.1, -0.1, 0], [-0.1, 0.1, 0]]
    assert _unordered_close([c["r_co_o1"] for c in a["contacts"]], exp)
    assert _unordered_close([c["r_co_o2"] for c in a["contacts"]], exp)


def test_fixture_cylinder_box_and_wedge_known_answers(arrangements):
    a = arrangements["tests/cylinder_box"]  # test_parsing.py:114-146
    assert len(a["contacts"]) == 10
    for c in a["contacts"]:
        if c["object1_name"] == "ee":
            assert np.allclose(c["normal"], [0, 0, -1])
        else:
            assert np.allclose(np.abs(c["normal"]), [1, 0, 0])
    w = arrangements["tests/wedge_box"]  # test_parsing.py:149-182
    wedge = [b for b in w["bodies"] if b["name"] == "wedge"][0]
    assert np.allclose(wedge["com"], [-0.05, 0, 0.1]) and len(w["contacts"]) == 8
    cz = np.array([np.sin(np.pi / 4), 0, np.cos(np.pi / 4)])
    for c in w["contacts"]:
        assert np.allclose(c["normal"], [0, 0, -1] if c["object1_name"] == "ee" else -cz)


def test_fixture_counts_match_survey(arrangements):
    counts = {k: (len(v["bodies"]), len(v["contacts"])) for k, v in arrangements.items()}
    assert counts["pink_bottle"] == (1, 4) and counts["foam_die2"] == (2, 8) and counts["box_arch"] == (3, 16)
    assert counts["blue_cups"] == (7, 28) and counts["wedge"] == (2, 8)
    b = arrangements["pink_bottle"]["bodies"][0]
    assert np.isclose(b["mass"], 0.827) and np.allclose(b["com"], [0.035, -0.035, 0.1275])
    assert np.allclose(np.diag(b["inertia"]), [4.734575e-3, 4.734575e-3, 5.065375e-4])


# ---------------------------------------------------------------------------------------------------------
# numpy twin of contact_constraints.h:50-194 (written from the formulas of SURVEY.md section 8)
def np_object_dynamics(P, forces, C, w, al, a):
    nb = P.nb
    F = np.zeros((nb, 3)); T = np.zeros((nb, 3))
    m = P.body_params[:, 0]; com = P.body_params[:, 1:4] / m[:, None]
    for i in range(P.nc):
        f = forces[i] * P.contact_normal[i] if P.nf == 1 else forces[3 * i:3 * i + 3]
        b1, b2 = P.contact_body1[i], P.contact_body2[i]
        if b1 >= 0:
            F[b1] += f; T[b1] += np.cross(P.contact_r1[i] - com[b1], f)
        F[b2] -= f; T[b2] += np.cross(P.contact_r2[i] - com[b2], -f)
    ddC = (skew(al) + skew(w) @ skew(w)) @ C
    out = []
    for b in range(nb):
        v = P.body_params[b, 4:]
        I = np.array([[v[0], v[1], v[2]], [v[1], v[3], v[4]], [v[2], v[4], v[5]]])
        gif = m[b] * C.T @ (a + ddC @ com[b] - P.gravity)
        we, ae = C.T @ w, C.T @ al
        tau = np.cross(we, I @ we) + I @ ae
        out += [(gif - F[b]) / m[b], (tau - T[b]) / m[b]]
    return np.concatenate(out)


def np_friction_rows(P, forces):
    out = []
    for i in range(P.nc):
        f = forces[3 * i:3 * i + 3]
        fn = P.contact_normal[i] @ f; t0, t1 = P.contact_span[i] @ f; mu = P.contact_mu[i]
        out += [fn, mu * fn - t0 - t1, mu * fn - t0 + t1, mu * fn + t0 - t1, mu * fn + t0 + t1]
    return np.array(out)


def _rot(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q); x, y, z, s = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - s * z), 2 * (x * z + s * y)],
                     [2 * (x * y + s * z), 1 - 2 * (x * x + z * z), 2 * (y * z - s * x)],
                     [2 * (x * z - s * y), 2 * (y * z + s * x), 1 - 2 * (x * x + y * y)]])


@pytest.mark.parametrize("name", ["pink_bottle", "foam_die2", "box_arch", "blue_cups", "wedge", "tests/cylinder_box"])
@pytest.mark.parametrize("nf", [3, 1])
def test_oracle_matches_numpy_twin(arrangements, name, nf):
    P = thing_problem(arrangements[name], nf=nf)
    O = Oracle(P)
    rng = np.random.default_rng(7)
    for _ in range(5):
        f = rng.normal(size=nf * P.nc); C = _rot(rng); w, al, a = rng.normal(size=(3, 3))
        assert np.allclose(O.object_dynamics(f, C, w, al, a), np_object_dynamics(P, f, C, w, al, a), rtol=0, atol=1e-12 * 50)
        if nf == 3:
            assert np.allclose(O.friction_rows(f), np_friction_rows(P, f), rtol=0, atol=1e-13)


def test_static_equilibrium_and_free_fall(arrangements):
    P = thing_problem(arrangements["pink_bottle"])
    O = Oracle(P)
    m = P.body_params[0, 0]
    f = np.zeros(12); f[2::3] = -m * G / 4  # each contact carries a quarter of the weight
    z = np.zeros(3)
    assert np.abs(O.object_dynamics(f, np.eye(3), z, z, z)).max() < 1e-12
    assert np.all(O.friction_rows(f) > 0)
    assert np.abs(O.object_dynamics(np.zeros(12), np.eye(3), z, z, P.gravity)).max() < 1e-12  # free fall


def test_wedge_minimum_friction_known_answer(arrangements):
    """arrangements.yaml:184: mu_margin 0.1683 was chosen so that mu = 0.3 - 0.1683 = 0.1317 = tan(7.5 deg) is
    the smallest friction coefficient for which the 15-degree wedge arrangement can be balanced statically
    when the tray may tilt (compute_minimum_mu.py:75-110)."""
    P = thing_problem(arrangements["wedge"])
    O = Oracle(P)
    z = np.zeros(3)

    def feasible(mu, th):
        P.contact_mu[:] = mu
        Ot = Oracle(P)
        C = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
        n = 3 * P.nc
        A = np.zeros((6 * P.nb, n)); b0 = Ot.object_dynamics(np.zeros(n), C, z, z, z)
        E = np.zeros((5 * P.nc, n))
        for j in range(n):
            e = np.zeros(n); e[j] = 1
            A[:, j] = Ot.object_dynamics(e, C, z, z, z) - b0
            E[:, j] = Ot.friction_rows(e)
        r = linprog(np.zeros(n), A_ub=-E, b_ub=np.zeros(5 * P.nc), A_eq=A, b_eq=-b0, bounds=[(None, None)] * n, method="highs")
        return r.status == 0

    def mu_min(th):
        lo, hi = 0.0, 1.0
        for _ in range(22):
            mid = 0.5 * (lo + hi)
            lo, hi = (lo, mid) if feasible(mid, th) else (mid, hi)
        return hi

    best = min(mu_min(np.deg2rad(d)) for d in np.linspace(-10, 10, 41))  # includes +-7.5 deg
    assert abs(best - np.tan(np.deg2rad(7.5))) < 2e-3
    assert abs(best - (0.3 - 0.1683)) < 2e-3


# ---------------------------------------------------------------------------------------------------------
# kinematics and Jacobians
def test_tray_is_level_at_home(arrangements):
    """The chain model's documented premise (upright_amd/robots.py): at the reference's home configuration
    (robots/thing.yaml:16) the tray normal is vertical to within the calibration residual."""
    O = Oracle(thing_problem(arrangements["pink_bottle"]))
    k = O.ee_kinematics(np.concatenate([THING_HOME, np.zeros(18)]))
    C = k[3:12].reshape(3, 3)
    assert C[2, 2] > 0.9998 and np.allclose(C @ C.T, np.eye(3), atol=1e-12)


def test_kinematics_consistency_and_finite_differences(arrangements):
    P = thing_problem(arrangements["pink_bottle"])
    O = Oracle(P)
    rng = np.random.default_rng(3)
    x = np.concatenate([THING_HOME, np.zeros(18)]) + rng.uniform(-0.3, 0.3, 27)
    out, J = O.ee_kinematics(x, jac=True)
    h = 1e-6
    Jfd = np.stack([(O.ee_kinematics(x + h * e) - O.ee_kinematics(x - h * e)) / (2 * h) for e in np.eye(27)], axis=1)
    assert np.abs(J - Jfd).max() < 1e-7
    q, v, a = x[:9], x[9:18], x[18:]
    assert np.allclose(out[12:15], J[0:3, :9] @ v, atol=1e-12)                      # v = J_p qdot
    assert np.allclose(out[18:21], J[12:15, :9] @ v + J[12:15, 9:18] @ a, atol=1e-12)  # classical acceleration = dv/dt
    assert np.allclose(out[21:24], J[15:18, :9] @ v + J[15:18, 9:18] @ a, atol=1e-12)  # alpha = dw/dt
    Cd = (J[3:12, :9] @ v).reshape(3, 3); C = out[3:12].reshape(3, 3); S = Cd @ C.T
    assert np.allclose([S[2, 1], S[0, 2], S[1, 0]], out[15:18], atol=1e-12)          # dC/dt = S(w) C
    u = rng.uniform(-1, 1, P.nu)
    g, gx, gu = O.eq_constraint(x, u)
    gfd = np.stack([(O.eq_constraint(x + h * e, u, jac=False) - O.eq_constraint(x - h * e, u, jac=False)) / (2 * h) for e in np.eye(27)], axis=1)
    assert np.abs(gx - gfd).max() < 1e-6
    assert np.abs(gu[:, :9]).max() == 0.0  # jerk does not enter the balancing constraint
    # normalisation 1/sqrt(6 nb) (balancing_constraints.cpp:144-151) on top of the core function
    k = O.ee_kinematics(x)
    core = O.object_dynamics(u[9:], k[3:12].reshape(3, 3), k[15:18], k[21:24], k[18:21])
    assert np.allclose(g, core / np.sqrt(6.0), atol=1e-13)


def test_stage_cost_and_terminal_constraint(arrangements):
    P = thing_problem(arrangements["pink_bottle"])
    O = Oracle(P)
    rng = np.random.default_rng(4)
    x = np.concatenate([THING_HOME, np.zeros(18)]) + rng.uniform(-0.2, 0.2, 27); u = rng.uniform(-1, 1, P.nu)
    c, gx, gu, H, R = O.stage_cost(0.3, x, u)
    h = 1e-6
    gfd = np.array([(O.stage_cost(0.3, x + h * e, u, derivs=False) - O.stage_cost(0.3, x - h * e, u, derivs=False)) / (2 * h) for e in np.eye(27)])
    assert np.abs(gx - gfd).max() < 1e-7 and np.allclose(gu, P.Rdiag * u)
    p = O.ee_kinematics(x)[:3]
    e = p - P.way_p[0]
    assert np.isclose(c, 0.5 * np.sum(P.Qdiag * x ** 2) + 0.5 * np.sum(P.Rdiag * u ** 2) + 0.5 * e @ e)
    Jp = O.ee_kinematics(x, jac=True)[1][:3]
    assert np.allclose(H, np.diag(P.Qdiag) + Jp.T @ Jp, atol=1e-12)  # Gauss-Newton, end_effector_cost.h:80-82
    cN, CN = O.terminal_constraint(2.0, x)
    assert np.allclose(cN, np.concatenate([P.way_p[0] - p, x[9:]])) and np.allclose(CN[:3], -Jp) and np.allclose(CN[3:, 9:], np.eye(18))
    xn = O.dynamics(x, u)
    dt = P.dt; q, v, a, j = x[:9], x[9:18], x[18:], u[:9]
    assert np.allclose(xn, np.concatenate([q + dt * v + dt ** 2 / 2 * a + dt ** 3 / 6 * j, v + dt * a + dt ** 2 / 2 * j, a + dt * j]))


# ---------------------------------------------------------------------------------------------------------
# QP and SQP
def test_qp_against_independent_solver(arrangements):
    """Horizon-2 QP of the first SQP iteration solved by scipy's SLSQP (independent algorithm): the strictly
    convex QP has one minimiser."""
    P = thing_problem(arrangements["pink_bottle"], N=2, terminal_constraint=False)
    O = Oracle(P)
    x0 = level_tray_states(1, seed=5)[0]
    P.way_p = waypoints_for(P, x0[None])[0]
    O = Oracle(P)
    xs, us = stationary_guess(x0, P.N, P.nu); xs, us = xs[0], us[0]
    dxo, duo, st, rc = O.qp_step(0.0, x0, xs, us)
    assert rc == 0 and max(st.qp_res) < 1e-8
    nx, nu, N, h = P.nx, P.nu, P.N, P.dt
    nz = N * nx + N * nu
    ix = lambda k: slice((k - 1) * nx, k * nx)
    iu = lambda k: slice(N * nx + k * nu, N * nx + (k + 1) * nu)
    Hm = np.zeros((nz, nz)); gv = np.zeros(nz); Aeq = []; beq = []; G = []; d = []
    A = np.eye(nx); A[:9, 9:18] = h * np.eye(9); A[:9, 18:] = h * h / 2 * np.eye(9); A[9:18, 18:] = h * np.eye(9)
    Bm = np.zeros((nx, nu)); Bm[:9, :9] = h ** 3 / 6 * np.eye(9); Bm[9:18, :9] = h * h / 2 * np.eye(9); Bm[18:, :9] = h * np.eye(9)
    E = np.stack([O.ineq_constraint(e) for e in np.eye(nu)], axis=1)
    for k in range(N):
        c, gx, gu, Hxx, Rd = O.stage_cost(k * h, xs[k], us[k])
        if k >= 1:
            Hm[ix(k), ix(k)] += h * Hxx; gv[ix(k)] += h * gx
        Hm[iu(k), iu(k)] += h * np.diag(Rd); gv[iu(k)] += h * gu
        g, Gx, Gu = O.eq_constraint(xs[k], us[k])
        r = np.zeros((6, nz)); r[:, iu(k)] = Gu
        if k >= 1: r[:, ix(k)] = Gx
        Aeq.append(r); beq.append(-g)
        r = np.zeros((nx, nz)); r[:, ix(k + 1)] = -np.eye(nx); r[:, iu(k)] = Bm
        if k >= 1: r[:, ix(k)] = A
        Aeq.append(r); beq.append(-(O.dynamics(xs[k], us[k]) - xs[k + 1]))
        r = np.zeros((20, nz)); r[:, iu(k)] = E; G.append(r); d.append(O.ineq_constraint(us[k]))
    Aeq, beq, G, d = np.vstack(Aeq), np.concatenate(beq), np.vstack(G), np.concatenate(d)
    lb = np.concatenate([np.concatenate([P.x_lb - xs[k] for k in range(1, N + 1)]), np.concatenate([P.u_lb - us[k] for k in range(N)])])
    ub = np.concatenate([np.concatenate([P.x_ub - xs[k] for k in range(1, N + 1)]), np.concatenate([P.u_ub - us[k] for k in range(N)])])
    zo = np.concatenate([dxo[1:].ravel(), duo.ravel()])
    f = lambda z: 0.5 * z @ Hm @ z + gv @ z
    res = minimize(f, np.zeros(nz), jac=lambda z: Hm @ z + gv, method="SLSQP", bounds=list(zip(lb, ub)),
                   constraints=[{"type": "eq", "fun": lambda z: Aeq @ z - beq, "jac": lambda z: Aeq},
                                {"type": "ineq", "fun": lambda z: G @ z + d, "jac": lambda z: G}],
                   options={"maxiter": 500, "ftol": 1e-14})
    assert np.abs(Aeq @ zo - beq).max() < 1e-8 and (G @ zo + d).min() > -1e-8
    assert f(zo) <= f(res.x) + 1e-7 * max(1, abs(f(res.x)))          # at least as good as the independent solve
    assert np.abs(zo - res.x).max() < 1e-3 * max(1, np.abs(zo).max())


def test_sqp_converges_to_feasible_trajectory(arrangements):
    P = thing_problem(arrangements["pink_bottle"], sqp_iters=15)
    x0 = level_tray_states(1, seed=9)[0]
    P.way_p = waypoints_for(P, x0[None])[0]
    O = Oracle(P)
    xs0, us0 = stationary_guess(x0, P.N, P.nu)
    xs, us, st, rc = O.solve(0.0, x0, xs0[0], us0[0])
    assert rc == 0 and st.sqp_iters_done < 15 and st.constraint_violation < 1e-4
    perf = O.performance(0.0, x0, xs, us)
    assert perf[1] < 1e-12 and perf[2] < 1e-8 and perf[3] < 1e-12   # defects, equalities, inequalities
    assert np.allclose(O.ee_kinematics(xs[-1])[:3], P.way_p[0], atol=1e-4) and np.abs(xs[-1, 9:]).max() < 1e-5
    h = np.concatenate([O.ineq_constraint(u) for u in us])
    assert h.min() > -1e-7


def test_qp_statistics_of_headline_sample(arrangements):
    """IPM behaviour on the benchmark distribution: every instance converges within the HPIPM cap (30)."""
    P = thing_problem(arrangements["pink_bottle"])
    x0 = level_tray_states(24, seed=0); way = waypoints_for(P, x0); xs0, us0 = stationary_guess(x0, P.N, P.nu)
    its = []
    for b in range(24):
        P.way_p = way[b]
        _, _, st, rc = Oracle(P).qp_step(0.0, x0[b], xs0[b], us0[b])
        assert rc == 0 and max(st.qp_res) < P.qp_tol
        its.append(st.qp_iters_last)
    assert max(its) <= 20


@pytest.mark.parametrize("name", ["pink_bottle", "box_arch", "robust_8corner"])
@pytest.mark.parametrize("nf", [3, 1])
def test_wrench_map_against_reference_grasp_matrix(arrangements, name, nf):
    """Second reference-held answer for a1 / a2: the contact-force -> body-wrench map as upright_robust/modelling.py:83-103
    states it (compute_grasp_matrix, imported by tests/golden/make_fixtures.py: signs, body order, object 1 = "ee"
    skipped), independent of the C++ headers.  The oracle's d(object_dynamics)/d(forces) must equal
    -G / (m sqrt(6 nb)) with the torque rows shifted to each body's centre of mass."""
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from kkt_check import force_jacobian, force_jacobian_from_grasp

    arr = arrangements[name]
    gr = json.load(open(Path(__file__).resolve().parent / "golden" / "grasp.json"))[name]
    assert gr["names"] == [b["name"] for b in arr["bodies"]]
    P = thing_problem(arr, nf=nf)
    masses = [b["mass"] for b in arr["bodies"]]; coms = [b["com"] for b in arr["bodies"]]
    normals = [c["normal"] for c in arr["contacts"]] if nf == 1 else None
    D_ref = force_jacobian_from_grasp(gr["G"], masses, coms, P.nb, normals)
    O = Oracle(P)
    rng = np.random.default_rng(11)
    x = np.concatenate([THING_HOME, np.zeros(18)]) + rng.uniform(-0.1, 0.1, 27)
    u = rng.uniform(-1, 1, P.nu)
    gu = O.eq_constraint(x, u)[2]
    assert gu.shape == (6 * P.nb, P.nu)
    assert np.abs(gu[:, 9:] - D_ref).max() < 1e-13 * max(1.0, np.abs(D_ref).max())
    assert np.abs(gu[:, :9]).max() == 0.0                      # the jerk does not enter the rows
    assert np.abs(force_jacobian(P, P.body_params) - D_ref).max() < 1e-13 * max(1.0, np.abs(D_ref).max())   # the numpy twin of tests/kkt_check.py
    # the rows are affine in the forces: g(u + df) - g(u) = D df exactly (to rounding)
    df = rng.uniform(-1, 1, P.nu - 9)
    u2 = u.copy(); u2[9:] += df
    g0 = O.eq_constraint(x, u, jac=False); g1 = O.eq_constraint(x, u2, jac=False)
    assert np.abs((g1 - g0) - D_ref @ df).max() < 1e-12 * max(1.0, np.abs(D_ref).max())


def _inertial_fixture():
    return json.load(open(Path(__file__).resolve().parent / "golden" / "inertial.json"))


def inertial_wrench_about_ee_origin(P, arr, rows):
    """m [force rows; torque rows + c x force rows] per body: the inertial wrench about the EE origin out of the per-body
    residual of contact_constraints.h:79-101 (which is divided by m and takes its torque about the centre of mass)."""
    out = []
    for b, body in enumerate(arr["bodies"]):
        m, c = body["mass"], np.asarray(body["com"])
        F = m * rows[6 * b:6 * b + 3]
        T = m * rows[6 * b + 3:6 * b + 6] + np.cross(c, F)
        out.append(np.concatenate([F, T]))
    return np.array(out)


@pytest.mark.parametrize("name", ["pink_bottle", "box_arch", "robust_8corner"])
def test_inertial_half_against_reference_spatial_mass_matrix(arrangements, name):
    """Third reference-held answer (a1 / a3): the inertial half of the object-dynamics residual as upright_robust states it in
    numpy, independent of the C++ headers -- UncertainObject.M (modelling.py:47-77: spatial mass matrix about the EE origin,
    parallel-axis inertia) and body_gravity6 (utils.py:5-13), both dumped by tests/golden/make_fixtures.py.  With zero contact
    forces and omega = 0 the residual of body b is the wrench M_b (A - G) with A = [C_ew a; C_ew alpha].
    omega != 0 stays unpinned: the bias term needs rigeo.skew6, which is absent here."""
    fx = _inertial_fixture()
    arr = arrangements[name]
    Ms = np.asarray(fx["arrangements"][name]["M"])
    assert fx["arrangements"][name]["names"] == [b["name"] for b in arr["bodies"]]
    P = thing_problem(arr)
    O = Oracle(P)
    rng = np.random.default_rng(5)
    z = np.zeros(3)
    for gcase in fx["gravity"]:
        C_ew = np.asarray(gcase["C_ew"]); G = np.asarray(gcase["G"])
        assert np.abs(G[:3] - C_ew @ np.asarray(P.gravity)).max() < 1e-14 and np.abs(G[3:]).max() == 0.0
        for _ in range(3):
            a, al = rng.normal(size=(2, 3)) * 3.0
            A = np.concatenate([C_ew @ a, C_ew @ al])
            rows = O.object_dynamics(np.zeros(3 * P.nc), C_ew.T, z, al, a)
            got = inertial_wrench_about_ee_origin(P, arr, rows)
            ref = np.stack([Ms[b] @ (A - G) for b in range(P.nb)])
            assert np.abs(got - ref).max() < 1e-12 * max(1.0, np.abs(ref).max()), (name, np.abs(got - ref).max())


@pytest.mark.parametrize("name", ["pink_bottle", "box_arch"])
def test_friction_rows_against_reference_cone_generators(arrangements, name):
    """Fourth reference-held answer (a4): upright_robust/modelling.py:34-44 states every contact's linearised friction cone in SPAN
    form, the four generators normal +- mu span_i (dumped by tests/golden/make_fixtures.py next to the grasp matrix).  The cone
    they span is the face form of contact_constraints.h:50-77, so the oracle's five rows on generator g_i must read
    (1, then mu -+ t_0 -+ t_1) = (1, 0, 0, 2 mu, 2 mu) up to the order of the facets: non-negative, exactly two facets active,
    the other two at 2 mu -- on every contact; positive combinations stay inside, a step beyond a facet leaves."""
    gr = json.load(open(Path(__file__).resolve().parent / "golden" / "grasp.json"))[name]
    arr = arrangements[name]
    P = thing_problem(arr)
    O = Oracle(P)
    rng = np.random.default_rng(3)
    for ci, (S, mu) in enumerate(zip(gr["S"], gr["mu"])):
        S = np.asarray(S)                       # 3 x 4
        assert abs(mu - arr["contacts"][ci]["mu"]) < 1e-15
        for gi in range(4):
            f = np.zeros(3 * P.nc); f[3 * ci:3 * ci + 3] = S[:, gi]
            rows = O.friction_rows(f)[5 * ci:5 * ci + 5]
            assert abs(rows[0] - 1.0) < 1e-14                                   # normal component of a generator
            srt = np.sort(rows[1:])
            assert np.abs(srt - np.array([0.0, 0.0, 2 * mu, 2 * mu])).max() < 1e-14, (ci, gi, rows)
        z = rng.uniform(0.1, 1.0, 4)
        f = np.zeros(3 * P.nc); f[3 * ci:3 * ci + 3] = S @ z
        assert O.friction_rows(f)[5 * ci:5 * ci + 5].min() > 0.0               # the interior of the span is inside the faces
        f[3 * ci:3 * ci + 3] = S @ np.array([1.0, 1.0, -0.2, 0.0])
        assert O.friction_rows(f)[5 * ci:5 * ci + 5].min() < 0.0               # outside the span: a facet is violated


def test_oracle_minimiser_does_not_depend_on_the_centrality_safeguard(arrangements):
    """The two oracle builds (with and without the step-length safeguard of round 5, oracle/Makefile) reach the same minimiser of the
    same headline QPs along different paths -- the CPU half of test_converged_qps_do_not_depend_on_the_interior_point_path (why
    qp_tol = 1e-12 with the cap of 30: see there)."""
    P = thing_problem(arrangements["pink_bottle"], qp_tol=1e-12, qp_iter_max=30)
    x0 = level_tray_states(6, seed=7)
    way = waypoints_for(P, x0)
    xs0, us0 = stationary_guess(x0, P.N, P.nu)
    paths_differ = False
    for b in range(6):
        P.way_p = way[b]
        dxa, dua, sa, rca = Oracle(P).qp_step(0.0, x0[b], xs0[b], us0[b])
        dxb, dub, sb, rcb = Oracle(P, variant="ngam0").qp_step(0.0, x0[b], xs0[b], us0[b])
        assert rca in (0, 1) and rcb in (0, 1)
        assert np.abs(dxa - dxb).max() < 2e-5 * max(1, np.abs(dxb).max()) and np.abs(dua - dub).max() < 2e-5 * max(1, np.abs(dub).max())
        paths_differ |= not np.array_equal(dxa, dxb)
    assert paths_differ     # (bitwise different results: the second library really runs another rule)
