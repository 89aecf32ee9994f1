"""Independent check of the optimality conditions of the SQP sub-problem.

The QP is assembled HERE, in numpy, from the per-knot linearisation records the engine produced
(`BatchMPC.lin_records()`) and the problem constants -- never through the oracle's interior-point code and never
through the kernels' own residual evaluation.  Given the primal point and the multipliers a QP kernel ended with
(`BatchMPC.qp_kkt()`), `kkt_residuals` returns the four residuals of the KKT system of

    min   sum_{k<N} h [ 1/2 (x_k - xd)' Q (x_k - xd) + 1/2 u_k' R u_k + ee_k(q_k) ]
    s.t.  x_{k+1} = A x_k + B u_k,  x_0 given                                    (pi_{k+1})
          g_k + C_k (x_k - xs_k) + Df (f_k - fs_k) = 0      [ = nu_k / Z if the rows are softened ]   (nu_k)
          [p_d - p - Jp (q_N - qs_N); v_N; a_N] = 0                              (yN)
          x_lb <= x_k <= x_ub (k >= 1), u_lb <= u_k <= u_ub, friction pyramid rows E f_k >= 0,
          d_r + J_r (q_k - qs_k) >= 0 (k = 1 .. N-1)                             (lam >= 0)

ee_k is the Gauss-Newton model of the end-effector cost at the linearisation point (value, gradient, J'WJ of the record).
A convex QP: a primal-dual point with zero residuals IS its solution, whatever produced it.
"""
import numpy as np


def dynamics_matrices(nq, h):
    """Exact discretisation of the triple integrator (system_dynamics.h:15-22)."""
    I, Z = np.eye(nq), np.zeros((nq, nq))
    A = np.block([[I, h * I, 0.5 * h * h * I], [Z, I, h * I], [Z, Z, I]])
    B = np.vstack([h ** 3 / 6.0 * I, 0.5 * h * h * I, h * I])
    return A, B


def friction_rows(P):
    """(5 nc) x (3 nc): rows [n.f; mu n.f -+ t0 -+ t1] per contact (contact_constraints.h:56-75), t = S f."""
    E = np.zeros((5 * P.nc, 3 * P.nc))
    for c in range(P.nc):
        n, S, mu = np.asarray(P.contact_normal[c]), np.asarray(P.contact_span[c]).reshape(2, 3), float(P.contact_mu[c])
        rows = [n, mu * n - S[0] - S[1], mu * n - S[0] + S[1], mu * n + S[0] - S[1], mu * n + S[0] + S[1]]
        E[5 * c:5 * c + 5, 3 * c:3 * c + 3] = rows
    return E


def force_jacobian(P, body_params):
    """d(object_dynamics)/d(forces), (6 nb) x (nf nc): every contact force acts on object 2 with a minus sign and on
    object 1 (when it is a balanced body) with a plus sign; residual = (inertial - contact wrench) / m / sqrt(6 nb)
    (contact_constraints.h:107-157,96-101; balancing_constraints.cpp:144-151)."""
    nb, nc, nf = P.nb, P.nc, P.nf
    D = np.zeros((6 * nb, nf * nc))
    scale = 1.0 / np.sqrt(6.0 * nb)
    for c in range(nc):
        dirs = np.eye(3) if nf == 3 else np.asarray(P.contact_normal[c]).reshape(1, 3)
        for a, f in enumerate(dirs):
            col = nf * c + a
            for body, r, sgn in ((int(P.contact_body1[c]), P.contact_r1[c], 1.0), (int(P.contact_body2[c]), P.contact_r2[c], -1.0)):
                if body < 0:
                    continue
                m = body_params[body][0]
                com = np.asarray(body_params[body][1:4]) / m
                F = sgn * f
                T = np.cross(np.asarray(r) - com, F)
                D[6 * body:6 * body + 3, col] += -scale * F / m
                D[6 * body + 3:6 * body + 6, col] += -scale * T / m
    return D


def kkt_residuals(P, body_params, x0, xs, us, lin, sol, lin_layout=None):
    """Residuals [stationarity, equality, inequality (primal and dual feasibility), complementarity] (max norms) of one
    instance.  xs[N+1][nx], us[N][nu]: linearisation trajectory; lin[N+1][stride]: its records; sol: dict with dx, du,
    pi, nu, yN, lam of that instance (BatchMPC.qp_kkt() sliced)."""
    nq, nx, nu, N, h = P.nq, P.nx, P.nu, P.N, P.dt
    ne, nfc = 6 * P.nb, P.nf * P.nc
    npoly = 5 * P.nc if P.nf == 3 else 0
    no = len(P.pair_a) + len(P.proj_sph)
    nh = nq * (nq + 1) // 2
    o_g, o_gx = 0, ne
    o_cost = o_gx + ne * nx
    o_grad, o_hess = o_cost + 1, o_cost + 1 + nq
    o_obs = o_hess + nh
    A, Bm = dynamics_matrices(nq, h)
    E = friction_rows(P) if npoly else np.zeros((0, nfc))
    Df = force_jacobian(P, body_params)
    X = xs + sol["dx"]; U = us + sol["du"]
    pi, nuv, yN, lam = sol["pi"], sol["nu"], sol["yN"], sol["lam"]
    soft = P.slacks or {}
    soft_eq = bool(soft.get("equality", soft.get("poly_ineq")))
    Zpen = float(soft.get("lower_L2_penalty", 100.0))
    iu = np.triu_indices(nq)

    def hess_of(rec):
        H = np.zeros((nq, nq)); H[iu] = rec[o_hess:o_hess + nh]
        return H + np.triu(H, 1).T

    def soft_row(c, lm, Zl, zl):
        """softened row c + sigma >= 0, sigma >= 0 (multiplier gam), cost 1/2 Z sigma^2 + z sigma: stationarity in
        sigma is Z sigma + z = lam + gam.  The slack is not exported: take the smallest sigma that keeps the row
        feasible and gam >= 0; what remains to be checked is gam sigma ~ 0 (returned) and lam (c + sigma) ~ 0."""
        sig = max(0.0, (lm - zl) / Zl if Zl > 0 else 0.0, -c)
        gam = Zl * sig + zl - lm
        return c + sig, abs(gam * sig)

    r_stat = r_eq = r_in = r_comp = 0.0
    r_eq = max(r_eq, np.abs(X[0] - x0).max())
    for k in range(N + 1):
        rec = lin[k]
        lk = lam[k]
        l_xlo, l_xhi = lk[:nx], lk[nx:2 * nx]
        l_ulo, l_uhi = lk[2 * nx:2 * nx + nu], lk[2 * nx + nu:2 * nx + 2 * nu]
        l_f = lk[2 * nx + 2 * nu:2 * nx + 2 * nu + npoly]
        l_o = lk[2 * nx + 2 * nu + npoly:]
        # ---- primal feasibility and complementarity of the rows of this knot
        rows = []   # (value, multiplier, softened?, Z, z)
        sx, su, sp = bool(soft.get("state_box")), bool(soft.get("input_box")), bool(soft.get("poly_ineq"))
        ZL, ZU = float(soft.get("lower_L2_penalty", 100.0)), float(soft.get("upper_L2_penalty", 100.0))
        zL, zU = float(soft.get("lower_L1_penalty", 0.0)), float(soft.get("upper_L1_penalty", 0.0))
        if k >= 1:
            rows += [(X[k][i] - P.x_lb[i], l_xlo[i], sx, ZL, zL) for i in range(nx)]
            rows += [(P.x_ub[i] - X[k][i], l_xhi[i], sx, ZU, zU) for i in range(nx)]
        if k < N:
            rows += [(U[k][i] - P.u_lb[i], l_ulo[i], su, ZL, zL) for i in range(nu)]
            rows += [(P.u_ub[i] - U[k][i], l_uhi[i], su, ZU, zU) for i in range(nu)]
            if npoly:
                cf = E @ U[k][nq:]
                rows += [(cf[r], l_f[r], sp, ZL, zL) for r in range(npoly)]
        Jo = np.zeros((0, nq))
        if no and 1 <= k < N:
            d0 = rec[o_obs:o_obs + no]; Jo = rec[o_obs + no:o_obs + no + no * nq].reshape(no, nq)
            co = d0 + Jo @ (X[k][:nq] - xs[k][:nq])
            rows += [(co[r], l_o[r], sp, ZL, zL) for r in range(no)]
        for c, lm, is_soft, Zl, zl in rows:
            ce, cs = soft_row(c, lm, Zl, zl) if is_soft else (c, 0.0)
            r_in = max(r_in, -min(ce, 0.0), -min(lm, 0.0))
            r_comp = max(r_comp, abs(ce * lm), cs)
        # ---- stationarity in x_k (k >= 1; x_0 is fixed)
        if k >= 1:
            g = -pi[k] - l_xlo + l_xhi
            if k < N:
                g = g + h * P.Qdiag * (X[k] - P.xd) + A.T @ pi[k + 1]
                g[:nq] += h * (rec[o_grad:o_grad + nq] + hess_of(rec) @ (X[k][:nq] - xs[k][:nq]))
                C = rec[o_gx:o_gx + ne * nx].reshape(ne, nx)
                g = g + C.T @ nuv[k]
                if no:
                    g[:nq] -= Jo.T @ l_o
            elif P.terminal_constraint:
                Jp = rec[o_hess:o_hess + 3 * nq].reshape(3, nq)
                g[:nq] -= Jp.T @ yN[:3]
                g[nq:] += yN[3:]
            r_stat = max(r_stat, np.abs(g).max())
        if k < N:
            # ---- stationarity in u_k
            gu = h * P.Rdiag * U[k] - l_ulo + l_uhi
            gu[:nq] += Bm.T @ pi[k + 1]
            gu[nq:] += Df.T @ nuv[k]
            if npoly:
                gu[nq:] -= E.T @ l_f
            r_stat = max(r_stat, np.abs(gu).max())
            # ---- dynamics and the object-dynamics rows
            r_eq = max(r_eq, np.abs(A @ X[k] + Bm @ U[k][:nq] - X[k + 1]).max())
            C = rec[o_gx:o_gx + ne * nx].reshape(ne, nx)
            e = rec[o_g:o_g + ne] + C @ (X[k] - xs[k]) + Df @ (U[k][nq:] - us[k][nq:])
            if soft_eq:
                e = e - nuv[k] / Zpen
            r_eq = max(r_eq, np.abs(e).max())
        elif P.terminal_constraint:
            Jp = rec[o_hess:o_hess + 3 * nq].reshape(3, nq)
            eN = np.concatenate([rec[o_grad:o_grad + 3] - Jp @ (X[N][:nq] - xs[N][:nq]), X[N][nq:]])
            r_eq = max(r_eq, np.abs(eN).max())
    return np.array([r_stat, r_eq, r_in, r_comp])


def force_jacobian_from_grasp(G, masses, coms, nb, contact_normals=None):
    """The same Jacobian out of the reference's GRASP MATRIX (tests/golden/grasp.json: upright_robust/modelling.py:83-103,
    contact forces -> body wrenches about the END-EFFECTOR origin, +G1 on object 1 unless it is "ee", -G2 on object 2):
    d(object_dynamics)/d(forces) = -[G_F ; G_T - S(c_b) G_F] / (m_b sqrt(6 nb)) per body -- the torque rows shifted to the
    centre of mass (contact_constraints.h:126-154 takes moments about c) and the residual's scaling
    (contact_constraints.h:96-101, balancing_constraints.cpp:144-151).  contact_normals: frictionless problems (one
    normal-force coordinate per contact, contact_constraints.h:111-120)."""
    G = np.asarray(G, dtype=float)
    D = np.zeros_like(G)
    scale = 1.0 / np.sqrt(6.0 * nb)
    for b in range(nb):
        c = np.asarray(coms[b], dtype=float)
        S = np.array([[0, -c[2], c[1]], [c[2], 0, -c[0]], [-c[1], c[0], 0]])
        GF, GT = G[6 * b:6 * b + 3], G[6 * b + 3:6 * b + 6]
        D[6 * b:6 * b + 3] = -scale * GF / masses[b]
        D[6 * b + 3:6 * b + 6] = -scale * (GT - S @ GF) / masses[b]
    if contact_normals is not None:
        D = np.stack([D[:, 3 * i:3 * i + 3] @ np.asarray(n, dtype=float) for i, n in enumerate(contact_normals)], axis=1)
    return D
