"""Solver-level queries of the reference's controller interface that are answered from the LAST QP of the SQP:
`valueFunction`, `valueFunctionStateDerivative`, `stateInputEqualityConstraintLagrangian`
(/root/reference/upright_control/src/pybindings.cpp:398-412 -> ocs2::PythonInterface -> the solver's getValueFunction /
getStateInputEqualityConstraintLagrangian; ocs2_sqp answers the first with the Riccati cost-to-go of its last QP, HPIPM's
barrier-augmented P_k [UPSTREAM, absent: restated]).

The engine's QP kernels keep the cost-to-go matrices of their Riccati sweep in registers and do not store them.  What they do export
(upr_batch_qp_kkt / upr_batch_qp_slacks) is the primal-dual point the last QP ended at: costates pi_k, multipliers nu_k of the
object-dynamics rows, multipliers lam and slacks t of every inequality row.  From those this module rebuilds, in numpy on the host
(one instance, N = 20: 21 solves of a (nu + ne)-square system), exactly what the sweep factors:

    stage Hessians   Hxx_k = h Q + h Hee_k + diag(w_x) + Jo' diag(w_o) Jo      w = lam / t: the barrier weights of the last iterate
                     Huu_k = h R + diag(w_u) + E' diag(w_f) E                  (E: friction rows, force block)
    equality rows    C_k dx + [0 Df] du = 0   (hard; rho = 1e-6 on the dual block where the forces cannot span the rows, upr_qp.h),
                     or the penalty Z/2 |C dx + Df df|^2 where the rows are softened (upr_problem::soft_eq)
    terminal         P_N = diag(w_x) + (1 / rho_N) C_N' C_N                    (proximal terminal equality, rho_N = 1e-6)
    recursion        P_k = Hxx + A' P+ A - G' M^-1 G,   M = [[Huu + B' P+ B, D'], [D, -rho I]],   G = [B' P+ A; C]

(softened INEQUALITY rows are not covered: their factored weight needs the slack's own barrier pair, which the kernels do not
export -- control_bindings.ControllerInterface refuses the queries for such problems) and the gradient of the cost-to-go at the
plan's own state is the costate pi_k.  Around the plan
    V(t, x) ~ J(t) + pi(t)' (x - x*(t)) + 1/2 (x - x*(t))' P(t) (x - x*(t)),     dV/dx = pi(t) + P(t) (x - x*(t))
with pi, P, x*, J interpolated linearly between the knots (ocs2 LinearInterpolation).  Checked against finite differences of the
QP's optimal value over the observed state (tests/test_gpu_parity.py::test_value_function_against_finite_differences).
"""
import numpy as np


def _dynamics(nq, h):
    I, Z = np.eye(nq), np.zeros((nq, nq))
    A = np.block([[I, h * I, 0.5 * h * h * I], [Z, I, h * I], [Z, Z, I]])
    B = np.vstack([h ** 3 / 6.0 * I, 0.5 * h * h * I, h * I])
    return A, B


def record_layout(P):
    """Offsets of a linearisation record (BatchMPC.lin_records): g, dg/dx, cost, gradient, packed Hessian, collision rows."""
    nq, nx, ne = P.nq, P.nx, 6 * P.nb
    nh = nq * (nq + 1) // 2
    o_gx = ne
    o_cost = o_gx + ne * nx
    return dict(g=0, gx=o_gx, cost=o_cost, grad=o_cost + 1, hess=o_cost + 1 + nq, obs=o_cost + 1 + nq + nh, nh=nh)


def riccati_value_function(P, xs, us, lin, sol, E, Df, rho_N=1e-6, rho_prox=1e-6):
    """Cost-to-go matrices P_k (k = 0 .. N) and gradients p_k of the QP one instance's last `qp_kkt` call solved.
    xs[N+1][nx], us[N][nu]: linearisation trajectory; lin[N+1][stride]: its records; sol: qp_kkt() sliced to the instance (dx, du,
    pi, nu, yN, lam, slack); E (np x nfc): friction rows; Df (ne x nfc): d(object dynamics)/d(forces).
    Returns (Pk[N+1][nx][nx], pk[N+1][nx], X[N+1][nx], U[N][nu])."""
    nq, nx, nu, N, h = P.nq, P.nx, P.nu, P.N, P.dt
    ne, nfc = 6 * P.nb, P.nf * P.nc
    npoly = E.shape[0]
    no = len(P.pair_a) + len(P.proj_sph)
    o = record_layout(P)
    A, Bq = _dynamics(nq, h)
    Bf = np.hstack([Bq, np.zeros((nx, nfc))])                      # forces do not enter the dynamics
    X = xs + sol["dx"]; U = us + sol["du"]
    lam, t = sol["lam"], sol["slack"]
    w = lam / t
    soft = P.slacks or {}
    soft_eq = bool(soft.get("equality", soft.get("poly_ineq")))
    Z = float(soft.get("lower_L2_penalty", 100.0))
    iu = np.triu_indices(nq)
    D = np.hstack([np.zeros((ne, nq)), Df])

    def hee(rec):
        H = np.zeros((nq, nq)); H[iu] = rec[o["hess"]:o["hess"] + o["nh"]]
        return H + np.triu(H, 1).T

    Pk = np.zeros((N + 1, nx, nx)); pk = np.zeros((N + 1, nx))
    # terminal knot: box rows + the proximal form of the terminal equality
    PN = np.diag(w[N][:nx] + w[N][nx:2 * nx])
    if P.terminal_constraint:
        Jp = lin[N][o["hess"]:o["hess"] + 3 * nq].reshape(3, nq)
        CN = np.zeros((3 + 2 * nq, nx)); CN[:3, :nq] = -Jp; CN[3:, nq:] = np.eye(2 * nq)
        PN = PN + CN.T @ CN / rho_N
    Pk[N] = PN; pk[N] = sol["pi"][N]
    for k in range(N - 1, -1, -1):
        rec = lin[k]
        wk = w[k]
        Hxx = h * np.diag(P.Qdiag).astype(float)
        Hxx[:nq, :nq] += h * hee(rec)
        if k >= 1:
            Hxx += np.diag(wk[:nx] + wk[nx:2 * nx])
            if no:
                Jo = rec[o["obs"] + no:o["obs"] + no + no * nq].reshape(no, nq)
                Hxx[:nq, :nq] += Jo.T @ (wk[2 * nx + 2 * nu + npoly:2 * nx + 2 * nu + npoly + no, None] * Jo)
        Huu = h * np.diag(P.Rdiag).astype(float) + np.diag(wk[2 * nx:2 * nx + nu] + wk[2 * nx + nu:2 * nx + 2 * nu])
        if npoly:
            Huu[nq:, nq:] += E.T @ (wk[2 * nx + 2 * nu:2 * nx + 2 * nu + npoly, None] * E)
        C = rec[o["gx"]:o["gx"] + ne * nx].reshape(ne, nx)
        Pn = Pk[k + 1]
        Hux = np.zeros((nu, nx))
        if soft_eq:                                                # penalty form: no dual block
            Hxx = Hxx + Z * C.T @ C; Huu = Huu + Z * D.T @ D; Hux = Hux + Z * D.T @ C
            M = Huu + Bf.T @ Pn @ Bf
            G = Hux + Bf.T @ Pn @ A
        else:
            rho = rho_prox if nfc < ne else 1e-12
            M = np.block([[Huu + Bf.T @ Pn @ Bf, D.T], [D, -rho * np.eye(ne)]])
            G = np.vstack([Bf.T @ Pn @ A, C])
        Pk[k] = Hxx + A.T @ Pn @ A - G.T @ np.linalg.solve(M, G)
        Pk[k] = 0.5 * (Pk[k] + Pk[k].T)
        if k >= 1:
            pk[k] = sol["pi"][k]
        else:
            # pi_0 is not a multiplier of the QP (x_0 is fixed): the gradient of the Lagrangian in x_0
            g = h * P.Qdiag * (X[0] - P.xd) + A.T @ sol["pi"][1] + C.T @ sol["nu"][0]
            g[:nq] += h * (rec[o["grad"]:o["grad"] + nq] + hee(rec) @ (X[0][:nq] - xs[0][:nq]))
            pk[0] = g
    return Pk, pk, X, U


def qp_objective(P, xs, lin, X, U):
    """Value of the QP's objective at (X, U) per knot (Gauss-Newton model of the end-effector cost at the linearisation point):
    stage[k], k < N.  Their tail sums are the cost-to-go J_k of the plan."""
    nq, N, h = P.nq, P.N, P.dt
    o = record_layout(P)
    iu = np.triu_indices(nq)
    stage = np.zeros(N + 1)
    for k in range(N):
        rec = lin[k]
        H = np.zeros((nq, nq)); H[iu] = rec[o["hess"]:o["hess"] + o["nh"]]; H = H + np.triu(H, 1).T
        dq = X[k][:nq] - xs[k][:nq]
        ee = rec[o["cost"]] + rec[o["grad"]:o["grad"] + nq] @ dq + 0.5 * dq @ H @ dq
        stage[k] = h * (0.5 * (X[k] - P.xd) @ (P.Qdiag * (X[k] - P.xd)) + 0.5 * U[k] @ (P.Rdiag * U[k]) + ee)
    return stage


class ValueFunction:
    """Value function of the last QP of a solved `BatchMPC` handle (one instance), interpolated over the plan's knots."""

    def __init__(self, mpc, inst=0, riccati=True):
        from .engine import core_friction_rows

        P = mpc.problem
        ts, xs_sol, us_sol = mpc.solution()
        # the QP at the plan the solve ended with: linearised there, solved once more (its step is ~ 0 at a converged plan) -- ONE SQP
        # iteration past the QP ocs2's getValueFunction describes (the last one its solve ran); INTEGRATION.md section 2.  The handle's
        # statistics and its dispatch order keep describing the solve: the extra QP's are discarded
        with mpc.preserved_stats():
            sol = {k: v[inst] for k, v in mpc.qp_kkt().items()}
        lin = mpc.lin_records()[inst]
        nfc = P.nf * P.nc
        E = core_friction_rows(P, np.eye(nfc)).T if P.nf == 3 else np.zeros((0, nfc))
        Df = mpc.eq_input_jacobian(inst)[:, P.nq:]
        self.P = P
        self.t = np.asarray(ts[inst], dtype=float)
        self.nu = sol["nu"]
        self.Pk = None
        if not riccati:      # the equality multipliers alone (they need no barrier weights)
            return
        self.Pk, self.pk, self.X, self.U = riccati_value_function(P, xs_sol[inst][:, :P.nx], us_sol[inst], lin, sol, E, Df)
        stage = qp_objective(P, xs_sol[inst][:, :P.nx], lin, self.X, self.U)
        self.J = np.array([stage[k:].sum() for k in range(P.N + 1)])   # cost-to-go of the plan from knot k

    def _seg(self, t):
        ts = self.t
        s = min(max((float(t) - ts[0]) / self.P.dt, 0.0), float(self.P.N))
        j = min(int(s), self.P.N - 1)
        return j, s - j

    def gradient(self, t, x):
        j, a = self._seg(t)
        x = np.asarray(x, dtype=float)[:self.P.nx]
        g0 = self.pk[j] + self.Pk[j] @ (x - self.X[j])
        g1 = self.pk[j + 1] + self.Pk[j + 1] @ (x - self.X[j + 1])
        return (1 - a) * g0 + a * g1

    def value(self, t, x):
        j, a = self._seg(t)
        x = np.asarray(x, dtype=float)[:self.P.nx]
        v = []
        for k in (j, j + 1):
            d = x - self.X[k]
            v.append(self.J[k] + self.pk[k] @ d + 0.5 * d @ self.Pk[k] @ d)
        return (1 - a) * v[0] + a * v[1]

    def equality_multiplier(self, t):
        """nu(t): multipliers of the object-dynamics rows, piecewise linear between the knots (the last knot carries none)."""
        j, a = self._seg(t)
        n1 = self.nu[min(j + 1, self.P.N - 1)]
        return (1 - a) * self.nu[j] + a * n1
