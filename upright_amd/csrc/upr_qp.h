// upr_qp.h -- the stage-wise QP of one SQP iteration, solved entirely inside one workgroup per instance.
//
// What it replaces: the HPIPM OCP-QP solve the reference reaches through ocs2_sqp / hpipm_catkin
// [UPSTREAM, absent] with the settings bound at upright_control/src/pybindings.cpp:160-213
// (iter_max 30, controller.yaml:65-72).  Same problem, re-derived for this OCP's structure:
//
//   * Mehrotra predictor-corrector primal-dual interior point, one Riccati factorisation and two
//     back-substitutions per iteration, single step length for primal and dual.
//   * dynamics are the exact discretisation of the triple integrator (system_dynamics.h:15-22):
//     A'XA, B'X are linear combinations of nq-blocks, never dense products.
//   * contact forces enter neither the dynamics nor the cost cross terms, so the input Hessian is
//     blockdiag(jerk nq x nq, contacts 3x3 ...).  The object-dynamics equality (6 nb rows, constant
//     d/du) is eliminated exactly through the Schur complement S = Df Hff^-1 Df' (+ rho_s I).
//   * friction-cone rows (contact_constraints.h:50-77) and input/state boxes
//     (controller_interface.cpp:157-169,330-357) are the inequality set; the pyramid rows touch one
//     contact each, so their barrier Hessian is 3x3 per contact.
//   * the terminal equality (stationary_desired_position_constraint.h:35-74) is a proximal penalty
//     with explicit multiplier, i.e. the KKT point reached is the exact one.
//
// All small inverses are formed explicitly (Gauss-Jordan on SPD blocks) so that every
// back-substitution is a mat-vec that the whole workgroup executes in parallel.
#pragma once
#include "upr_kin.h"

// initial point of the IPM: slacks t = max(c, THR), multipliers MU0 / t.  (3, 0.1) sits on a broad plateau of the
// iteration count (thr 2..5, mu0 0.1..0.2) for every BASELINE configuration: 10.2 IPM iterations per QP of the headline
// batch against 11.4 with (10, 1), 11 against 14 for config 3, 7.7 against 8.6 for config 4
#define UPR_QP_THR 3.0
#define UPR_QP_MU0 0.1
// floor of the complementarity target relative to the tolerance (keeps lam/t bounded while the other
// residuals converge)
#define UPR_QP_SIGMA_FLOOR 1e-2
// Centrality safeguard of the step length in the first iterations (round 5; the wide neighbourhood N_-inf(gamma) of infeasible
// path-following methods: S. Wright, Primal-Dual Interior-Point Methods, ch. 6): while the iterate is still far from the central
// path (it < NIT: the infeasible start) the step is shortened once, by NBT, if a complementarity product of the trial point lies
// below NGAM times their average.  Without it a few rows run ahead to the boundary in the first iterations and the later steps
// stall at alpha ~ 0.4: two of the 1024 headline instances needed 16 iterations and three 14, now none more than 12 at an
// unchanged mean (10.4) -- and a launch lasts as long as its slowest PAIR of instances.  One more pass over
// the rows in each of the first four iterations.  Same constants in the oracle.  -DUPR_QP_NGAM=0.0 builds the round-4 rule.
#ifndef UPR_QP_NGAM
#define UPR_QP_NGAM 0.02
#endif
#define UPR_QP_NBT 0.9
#define UPR_QP_NIT 4
#define UPR_QP_RHO_S 1e-12
// When the contact-force block cannot span the equality rows (frictionless arrangements: nf nc < 6 nb) the
// Schur complement S = Df Hff^-1 Df' is rank deficient: part of the equality constrains the state alone.  It is
// then treated by the proximal method of multipliers,  C dx + Df df + e = rho (nu+ - nu),  with a rho that keeps
// the Riccati recursion conditioned; the interior-point iterations double as the proximal iterations.
#define UPR_QP_RHO_S_PROX 1e-6
static inline UPR_HD double upr_qp_rho_s(int ne, int nfc) { return (nfc < ne) ? UPR_QP_RHO_S_PROX : UPR_QP_RHO_S; }
static inline UPR_HD double upr_qp_rho_prox(int ne, int nfc) { return (nfc < ne) ? UPR_QP_RHO_S_PROX : 0.0; }
// Softened stage equality (upr_problem::soft_eq): the HPIPM slack pair of a row with equal L2 penalties Z and no L1 term
// eliminates to the penalty Z/2 |C dz + e|^2, i.e. the regularised equality C dz + e = nu / Z.  In the Schur-complement
// form that is rho_s = 1 / Z with no proximal carry-over, and the row's residual is C dz + e - nu / Z.
static inline UPR_HD double upr_qp_rho_soft(const upr_problem* P) { return P->soft_eq ? 1.0 / P->soft_L2_lower : 0.0; }
static inline UPR_HD double upr_qp_rho_s(const upr_problem* P, int ne, int nfc) { return P->soft_eq ? upr_qp_rho_soft(P) : upr_qp_rho_s(ne, nfc); }
static inline UPR_HD double upr_qp_rho_prox(const upr_problem* P, int ne, int nfc) { return P->soft_eq ? 0.0 : upr_qp_rho_prox(ne, nfc); }
#define UPR_QP_RHO_N 1e-6

struct upr_qp_args {
    const upr_problem* P;
    upr_dims d;
    const double* xs;    // [B][N+1][nx]
    const double* us;    // [B][N][nu]
    const double* x0;    // [B][nx]
    const double* lin;   // [B][N+1][lin_stride]
    const double* Df;    // [B][ne][nfc] constant d(object_dynamics)/d(forces)
    double* ws;          // [B][ws_stride]
    double* stats;       // [B][UPR_NSTATS]
    double* prof;        // optional [B][16] per-phase cycle counters (debug), NULL = off
    // optional export of the multipliers at exit (upr_batch_qp_kkt; kernels that keep them in registers / LDS write them
    // here): per instance [pi (N+1) nx][nu N ne][yN neN][lam (N+1) ni_stage], lam in the slot layout of upr_ineq_active
    double* kkt = nullptr;
    int kkt_stride = 0;
    // optional: linear feedback gains of THIS QP written by the kernel at exit, fb[B][N][nu][nx] (ocs2 sign: u = bias + K x).
    // Kernels that support it spare the separate gather kernel its launch and its re-read of the factors; the host sets
    // it on the last QP of an advance only (upr_api.hip)
    double* fb = nullptr;
    // optional dispatch order: workgroup i solves instance order[i] (a permutation of 0..B-1).  The host passes the
    // instances sorted by the IPM iteration count of their previous QP, longest first (ranked inside the line-search launch, upr_linesearch.h order_out)
    const int* order = nullptr;
    // optional: the IPM iteration count of every instance as one byte, iter_key[B] (clamped to 255): what the ranking inside the
    // line-search launch sorts by.  A compact array so that every workgroup of that launch reads all B keys with coalesced
    // requests (B bytes; ranking on stats[.][1] touched one cache line per instance and workgroup: O(B^2) lines)
    unsigned char* iter_key = nullptr;
};
static UPR_HDI void upr_qp_store_key(const upr_qp_args& A, int b, int it) { if (A.iter_key) A.iter_key[b] = (unsigned char)(it < 0 ? 0 : (it > 255 ? 255 : it)); }
static UPR_HDI int upr_qp_instance(const upr_qp_args& A, int wg) { return A.order ? A.order[wg] : wg; }
// (round 5: the slacks t of the rows behind their multipliers, same slot layout -- the barrier weights lam / t of the last iterate are
// what the value function of the QP is built from, upright_amd/value_function.py)
static inline UPR_HD int upr_kkt_doubles(const upr_dims& d) { return (d.N + 1) * d.nx + d.N * d.ne + d.neN + 2 * (d.N + 1) * d.ni_stage; }

// LDS layout (doubles)
struct upr_qp_lds {
    int Pm, Tm, Kx, Hjj, Lji, Cm, SC, Df, Yf, Sm, Lsi, Hff;
    int pv, wv, hx, bk, Xk, Xn, Uk, dxk, duk, sxk, suk, huj, huf, ku0, uf0, ee, snu, nuv, gxs, gus, Wx, Wu;
    int tk, lk, sv, wq, gjr, gjc, grad, hess, ob, hob, sgk, tak, gak, red, misc, total;   // sgk/tak/gak: slack part of the stage's soft rows   // ob: collision rows [d no][J no*nq]; hob: their barrier Hessian (packed)
};
static inline UPR_HD upr_qp_lds upr_qp_lds_layout(const upr_dims& d, int nt) {
    upr_qp_lds L; int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 1) & ~1; return r; };
    L.Pm = take(d.nx * d.nx); L.Tm = take(d.nq * d.nx); L.Kx = take(d.nq * d.nx); L.Hjj = take(d.nq * d.nq); L.Lji = take(d.nq * d.nq);
    L.Cm = take(d.ne * d.nx); L.SC = take(d.ne * d.nx); L.Df = take(d.ne * d.nfc); L.Yf = take(d.nfc * d.ne);
    L.Sm = take(d.ne * d.ne); L.Lsi = take(d.ne * d.ne); L.Hff = take(9 * d.nc);
    L.pv = take(d.nx); L.wv = take(d.nx); L.hx = take(d.nx); L.bk = take(d.nx); L.Xk = take(d.nx); L.Xn = take(d.nx);
    L.Uk = take(d.nu); L.dxk = take(d.nx); L.duk = take(d.nu); L.sxk = take(d.nx); L.suk = take(d.nu);
    L.huj = take(d.nq); L.huf = take(d.nfc); L.ku0 = take(d.nq); L.uf0 = take(d.nfc);
    L.ee = take(d.ne); L.snu = take(d.ne); L.nuv = take(d.ne); L.gxs = take(d.nx); L.gus = take(d.nu);
    L.Wx = take(d.nx); L.Wu = take(d.nu);
    L.tk = take(d.ni_stage); L.lk = take(d.ni_stage); L.sv = take(d.ni_stage); L.wq = take(d.ni_stage);
    int gm = d.nq > d.ne ? d.nq : d.ne;
    L.gjr = take(gm); L.gjc = take(gm); L.grad = take(d.nq); L.hess = take(d.nq * (d.nq + 1) / 2 > 3 * d.nq ? d.nq * (d.nq + 1) / 2 : 3 * d.nq);
    L.ob = take(d.no * (1 + d.nq)); L.hob = take(d.no > 0 ? d.nq * (d.nq + 1) / 2 : 0);
    L.sgk = take(d.soft ? d.ni_stage : 0); L.tak = take(d.soft ? d.ni_stage : 0); L.gak = take(d.soft ? d.ni_stage : 0);
    L.red = take(nt); L.misc = take(16);
    L.total = o;
    return L;
}

// ---- workgroup helpers ---------------------------------------------------------------------------
static inline UPR_HD double upr_reduce(const upr_ctx& ctx, double* red, double v, int op /*0 sum 1 max 2 min*/) {
    red[ctx.tid] = v;
    UPR_SYNC();
    for (int s = 1; s < ctx.nt; s <<= 1) {
        if ((ctx.tid & (2 * s - 1)) == 0 && ctx.tid + s < ctx.nt) {
            double a = red[ctx.tid], b = red[ctx.tid + s];
            red[ctx.tid] = (op == 0) ? a + b : (op == 1 ? (a > b ? a : b) : (a < b ? a : b));
        }
        UPR_SYNC();
    }
    double r = red[0];
    UPR_SYNC();
    return r;
}

// Cholesky factor of an SPD n x n matrix in LDS, in place in the lower triangle (right-looking), then
// its explicit inverse Li = L^-1 (lower, dense storage n x n with zeros above the diagonal).
// Applying L^-1 and L^-T as two triangular mat-vecs keeps the residual of every solve at
// eps * sqrt(cond), which an explicit M^-1 (eps * cond) does not -- the barrier weights reach 1e10.
static inline UPR_HD void upr_chol_inv(const upr_ctx& ctx, double* M, double* Li, int n, double* flag) {
    for (int p = 0; p < n; ++p) {
        const double piv = M[p * n + p];
        if (ctx.tid == 0 && !(piv > 0.0)) flag[0] = 1.0;
        const double dg = sqrt(piv > 0.0 ? piv : 1.0), idg = 1.0 / dg;
        UPR_SYNC();
        UPR_FOR(i, n) { if (i > p) M[i * n + p] *= idg; else if (i == p) M[p * n + p] = dg; }
        UPR_SYNC();
        const int m = n - p - 1;
        UPR_FOR(e, m * m) {
            int i = p + 1 + e / m, j2 = p + 1 + e % m;
            if (j2 <= i) M[i * n + j2] -= M[i * n + p] * M[j2 * n + p];
        }
        UPR_SYNC();
    }
    // column j of L^-1 by forward substitution; columns are independent
    UPR_FOR(j2, n) {
        for (int i = 0; i < j2; ++i) Li[i * n + j2] = 0.0;
        Li[j2 * n + j2] = 1.0 / M[j2 * n + j2];
        for (int i = j2 + 1; i < n; ++i) {
            double v = 0.0;
            for (int k = j2; k < i; ++k) v += M[i * n + k] * Li[k * n + j2];
            Li[i * n + j2] = -v / M[i * n + i];
        }
    }
    UPR_SYNC();
}

// 3x3 SPD block: m <- L^-1 (lower triangular, row-major 3x3, upper entries 0). false if not PD.
static inline UPR_HD bool upr_chol_inv3(double* m) {
    const double a = m[0], b = m[3], c = m[6], e = m[4], f = m[7], i = m[8];
    if (!(a > 0.0)) return false;
    const double n00 = upr_rsqrt(a), l10 = b * n00, l20 = c * n00;
    const double d1 = e - l10 * l10;
    if (!(d1 > 0.0)) return false;
    const double n11 = upr_rsqrt(d1), l21 = (f - l20 * l10) * n11;
    const double d2 = i - l20 * l20 - l21 * l21;
    if (!(d2 > 0.0)) return false;
    const double n22 = upr_rsqrt(d2);
    const double n10 = -l10 * n00 * n11, n21 = -l21 * n11 * n22, n20 = -(l20 * n00 + l21 * n10) * n22;
    m[0] = n00; m[1] = 0.0; m[2] = 0.0; m[3] = n10; m[4] = n11; m[5] = 0.0; m[6] = n20; m[7] = n21; m[8] = n22;
    return true;
}
// y = Lfi_blk * x (per contact block lower-triangular), and y = Lfi_blk' * x
static inline UPR_HD double upr_blk_lo(const upr_dims& d, const double* Lfi, const double* x, int i) {
    if (d.nf == 3) { int ci = i / 3, a = i % 3; const double* B = Lfi + 9 * ci; double v = 0.0; for (int b2 = 0; b2 <= a; ++b2) v += B[3 * a + b2] * x[3 * ci + b2]; return v; }
    return Lfi[i] * x[i];
}
static inline UPR_HD double upr_blk_up(const upr_dims& d, const double* Lfi, const double* x, int i) {
    if (d.nf == 3) { int ci = i / 3, a = i % 3; const double* B = Lfi + 9 * ci; double v = 0.0; for (int b2 = a; b2 < 3; ++b2) v += B[3 * b2 + a] * x[3 * ci + b2]; return v; }
    return Lfi[i] * x[i];
}

// structured dynamics: y = A' w  (in place allowed when y != w is not required: uses temporaries per j)
static inline UPR_HD void upr_At_vec(const upr_ctx& ctx, int nq, double h, const double* w, double* y) {
    const double h2 = 0.5 * h * h;
    UPR_FOR(j, nq) {
        double wq = w[j], wv = w[nq + j], wa = w[2 * nq + j];
        y[j] = wq; y[nq + j] = h * wq + wv; y[2 * nq + j] = h2 * wq + h * wv + wa;
    }
}
static inline UPR_HD double upr_Bt_vec_j(int nq, double h, const double* w, int j) {
    return (h * h * h / 6.0) * w[j] + (0.5 * h * h) * w[nq + j] + h * w[2 * nq + j];
}

// ---- inequality bookkeeping ------------------------------------------------------------------------
// slot j of stage k: is it a live inequality?   Slot layout: [x lo nx][x hi nx][u lo nu][u hi nu][friction np][collision no]
static inline UPR_HD bool upr_ineq_active(const upr_dims& d, int k, int j) {
    if (j < 2 * d.nx) return k >= 1;
    if (j >= 2 * d.nx + 2 * d.nu + d.np) return k >= 1 && k < d.N;   // collision rows: knots 1..N-1
    return k < d.N;
}
// soft rows (hpipm_interface SlackSettings): is slot j softened, and its L2 / L1 penalties
static inline UPR_HD bool upr_slot_soft(const upr_problem* P, const upr_dims& d, int j) {
    if (j < 2 * d.nx) return P->soft_state_box != 0;
    if (j < 2 * d.nx + 2 * d.nu) return P->soft_input_box != 0;
    return P->soft_poly != 0;
}
static inline UPR_HD bool upr_slot_upper(const upr_dims& d, int j) {
    return (j >= d.nx && j < 2 * d.nx) || (j >= 2 * d.nx + d.nu && j < 2 * d.nx + 2 * d.nu);
}
// One softened row c + sigma - t = 0 (lam), sigma - tau = 0 (gam), cost 1/2 Z sigma^2 + z sigma.  The slack step is
// eliminated from the Newton system: d sigma = -(a + w G dz) / D,  D = Z + w + w_s,
// a = (Z sigma + z - lam - gam) + (rc + lam rp) / t + (rc_s + gam rp_s) / tau.  What remains for the reduced system is
// the effective weight w (Z + w_s) / D and the gradient multiplier (rc + lam rp) / t - w a / D.
struct upr_soft_row {
    double Z, z, w, ws, D, a, rps;
};
static inline UPR_HD upr_soft_row upr_soft_terms(const upr_problem* P, const upr_dims& d, int j, double t, double lam, double sig, double tau, double gam,
                                                 double rp, double rc, double rcs) {
    upr_soft_row r;
    const bool up = upr_slot_upper(d, j);
    r.Z = up ? P->soft_L2_upper : P->soft_L2_lower; r.z = up ? P->soft_L1_upper : P->soft_L1_lower;
    r.w = lam / t; r.ws = gam / tau; r.D = r.Z + r.w + r.ws; r.rps = sig - tau;
    r.a = (r.Z * sig + r.z - lam - gam) + (rc + lam * rp) / t + (rcs + gam * r.rps) / tau;
    return r;
}

// value c_j at absolute (X, U); collision rows are linear in the step from the linearisation point:
// ob = [d no][J no*nq] of this knot, dxq = dx_k[0:nq]
static inline UPR_HD double upr_ineq_value(const upr_problem* P, const upr_dims& d, int j, const double* X, const double* U,
                                           const double* ob, const double* dxq) {
    if (j < d.nx) return X[j] - P->x_lb[j];
    j -= d.nx;
    if (j < d.nx) return P->x_ub[j] - X[j];
    j -= d.nx;
    if (j < d.nu) return U[j] - P->u_lb[j];
    j -= d.nu;
    if (j < d.nu) return P->u_ub[j] - U[j];
    j -= d.nu;
    if (j < d.np) {
        int ci = j / 5, r = j % 5;
        double e[3];
        upr_friction_row_jac(P, ci, r, e);
        const double* f = U + d.nq + 3 * ci;
        return e[0] * f[0] + e[1] * f[1] + e[2] * f[2];
    }
    j -= d.np;
    double v = ob[j];
    for (int i = 0; i < d.nq; ++i) v += ob[d.no + j * d.nq + i] * dxq[i];
    return v;
}
// G_j . (sx, su)
static inline UPR_HD double upr_ineq_dir(const upr_problem* P, const upr_dims& d, int j, const double* sx, const double* su, const double* ob) {
    if (j < d.nx) return sx[j];
    j -= d.nx;
    if (j < d.nx) return -sx[j];
    j -= d.nx;
    if (j < d.nu) return su[j];
    j -= d.nu;
    if (j < d.nu) return -su[j];
    j -= d.nu;
    if (j < d.np) {
        int ci = j / 5, r = j % 5;
        double e[3];
        upr_friction_row_jac(P, ci, r, e);
        const double* f = su + d.nq + 3 * ci;
        return e[0] * f[0] + e[1] * f[1] + e[2] * f[2];
    }
    j -= d.np;
    double v = 0.0;
    for (int i = 0; i < d.nq; ++i) v += ob[d.no + j * d.nq + i] * sx[i];
    return v;
}

struct upr_qp_state {
    upr_ctx ctx;
    const upr_problem* P;
    upr_dims d;
    upr_qp_lds o;
    double* L;          // LDS base
    const double* xs; const double* us; const double* x0; const double* lin; const double* Dfg;
    double* ws;
    double sigma_mu;    // corrector target
    int mode;           // 0 predictor, 1 corrector
};

// Load everything stage k needs into LDS: absolute iterate (Xk, Uk, Xn), dxk, duk, tk, lk, affine step
// (sxk, suk) and the linearisation record (Cm, ee <- g, grad, hess).  k == N loads the terminal data.
static inline UPR_HD void upr_qp_load_stage(upr_qp_state& S, int k) {
    const upr_ctx& ctx = S.ctx; const upr_dims& d = S.d; double* L = S.L; const upr_qp_lds& o = S.o;
    const double* dx = S.ws + d.ws_dx; const double* du = S.ws + d.ws_du;
    const double* sx = S.ws + d.ws_sx; const double* su = S.ws + d.ws_su;
    const double* rec = S.lin + (size_t)k * d.lin_stride;
    UPR_FOR(i, d.nx) {
        double dxi = dx[k * d.nx + i];
        L[o.dxk + i] = dxi;
        L[o.Xk + i] = S.xs[k * d.nx + i] + dxi;
        L[o.sxk + i] = sx[k * d.nx + i];
        if (k < d.N) L[o.Xn + i] = S.xs[(k + 1) * d.nx + i] + dx[(k + 1) * d.nx + i];
    }
    if (k < d.N) {
        UPR_FOR(i, d.nu) {
            double dui = du[k * d.nu + i];
            L[o.duk + i] = dui;
            L[o.Uk + i] = S.us[k * d.nu + i] + dui;
            L[o.suk + i] = su[k * d.nu + i];
        }
        UPR_FOR(i, d.ne * d.nx) L[o.Cm + i] = rec[d.lin_gx + i];
        UPR_FOR(i, d.ne) L[o.ee + i] = rec[d.lin_g + i];
    }
    if (d.no > 0 && k >= 1 && k < d.N) UPR_FOR(i, d.no * (1 + d.nq)) L[o.ob + i] = rec[d.lin_obs + i];
    UPR_FOR(i, d.nq) L[o.grad + i] = rec[d.lin_grad + i];
    {
        int nh = (k < d.N) ? d.nq * (d.nq + 1) / 2 : 3 * d.nq;
        UPR_FOR(i, nh) L[o.hess + i] = rec[d.lin_hess + i];
    }
    const double* t = S.ws + d.ws_t + (size_t)k * d.ni_stage;
    const double* lam = S.ws + d.ws_lam + (size_t)k * d.ni_stage;
    UPR_FOR(j, d.ni_stage) { L[o.tk + j] = t[j]; L[o.lk + j] = lam[j]; }
    if (d.soft) {
        const double* sg = S.ws + d.ws_sig + (size_t)k * d.ni_stage;
        const double* ta = S.ws + d.ws_tau + (size_t)k * d.ni_stage;
        const double* ga = S.ws + d.ws_gam + (size_t)k * d.ni_stage;
        UPR_FOR(j, d.ni_stage) { L[o.sgk + j] = sg[j]; L[o.tak + j] = ta[j]; L[o.gak + j] = ga[j]; }
    }
    UPR_SYNC();
}

// Per-stage assembly (after load_stage): barrier weights wq = lam/t, complementarity targets sv,
// reduced gradients gxs/gus (cost gradient + G's), barrier diagonals Wx/Wu, dynamics residual bk (k < N),
// equality residual ee (k < N).
static inline UPR_HD void upr_qp_assemble(upr_qp_state& S, int k) {
    const upr_ctx& ctx = S.ctx; const upr_dims& d = S.d; double* L = S.L; const upr_qp_lds& o = S.o;
    const upr_problem* P = S.P;
    const double h = P->dt;
    double* rcg = S.ws + d.ws_rc + (size_t)k * d.ni_stage;
    UPR_FOR(j, d.ni_stage) {
        double s = 0.0, w = 0.0;
        if (upr_ineq_active(d, k, j)) {
            double t = L[o.tk + j], lam = L[o.lk + j];
            double c = upr_ineq_value(P, d, j, L + o.Xk, L + o.Uk, L + o.ob, L + o.dxk);
            double rp = c - t;
            w = lam / t;
            // (the mode is uniform over the workgroup: it is tested OUTSIDE the per-lane soft / hard distinction -- with the
            // per-lane test outermost, lanes of a wave that mixes softened and hard rows lost their mode-2 value on gfx950)
            const bool is_soft = d.soft && upr_slot_soft(P, d, j);
            const double sig = is_soft ? L[o.sgk + j] : 0.0, tau = is_soft ? L[o.tak + j] : 1.0, gam = is_soft ? L[o.gak + j] : 0.0;
            rp += sig;
            if (S.mode == 2) s = -lam;                         // plain Lagrangian gradient
            else if (is_soft) {
                // softened row: the slack is eliminated, leaving an effective weight and gradient multiplier
                double* rcsg = S.ws + d.ws_rcs + (size_t)k * d.ni_stage;
                double rc = lam * t, rcs = gam * tau;
                if (S.mode == 1) {
                    const upr_soft_row a0 = upr_soft_terms(P, d, j, t, lam, sig, tau, gam, rp, rc, rcs);
                    const double gdz = upr_ineq_dir(P, d, j, L + o.sxk, L + o.suk, L + o.ob);
                    const double dsa = -(a0.a + a0.w * gdz) / a0.D;
                    const double dta = gdz + rp + dsa, dtaua = dsa + a0.rps;
                    const double dla = -lam - a0.w * dta, dga = -gam - a0.ws * dtaua;
                    rc = lam * t + dta * dla - S.sigma_mu; rcs = gam * tau + dtaua * dga - S.sigma_mu;
                    rcg[j] = rc; rcsg[j] = rcs;
                } else if (S.mode == 3) { rc = rcg[j]; rcs = rcsg[j]; }
                const upr_soft_row r = upr_soft_terms(P, d, j, t, lam, sig, tau, gam, rp, rc, rcs);
                s = (rc + lam * rp) / t - r.w * r.a / r.D - lam;
                w = r.w * (r.Z + r.ws) / r.D;
            }
            else if (S.mode == 0) s = w * rp;                  // predictor: rc = lam t
            else {
                double rc;
                if (S.mode == 1) {                             // corrector: build and keep the target
                    double dta = upr_ineq_dir(P, d, j, L + o.sxk, L + o.suk, L + o.ob) + rp;
                    double dla = -lam - w * dta;
                    rc = lam * t + dta * dla - S.sigma_mu;
                    rcg[j] = rc;
                } else rc = rcg[j];                            // mode 3: stored target
                s = (rc + lam * rp) / t - lam;
            }
        }
        L[o.sv + j] = s; L[o.wq + j] = w;
    }
    UPR_SYNC();
    const int ox = 0, ou = 2 * d.nx, op = 2 * d.nx + 2 * d.nu;
    UPR_FOR(i, d.nx) {
        double g = 0.0;
        if (k < d.N) {
            g = P->Qdiag[i] * (L[o.Xk + i] - P->xd[i]);
            if (i < d.nq) {
                double a = L[o.grad + i];
                for (int j = 0; j < d.nq; ++j) a += L[o.hess + upr_tri(d.nq, i, j)] * L[o.dxk + j];
                g += a;
            }
            g *= h;
        }
        if (d.no > 0 && i < d.nq && k >= 1 && k < d.N) {
            const int oc = 2 * d.nx + 2 * d.nu + d.np;
            for (int r = 0; r < d.no; ++r) g += L[o.ob + d.no + r * d.nq + i] * L[o.sv + oc + r];
        }
        L[o.gxs + i] = g + L[o.sv + ox + i] - L[o.sv + ox + d.nx + i];
        L[o.Wx + i] = L[o.wq + ox + i] + L[o.wq + ox + d.nx + i];
    }
    if (d.no > 0) UPR_FOR(e, d.nq * (d.nq + 1) / 2) {   // barrier Hessian of the collision rows: sum_r w_r J_r J_r'
        double v = 0.0;
        if (k >= 1 && k < d.N) {
            int i = 0, rem = e;
            while (rem >= d.nq - i) { rem -= d.nq - i; ++i; }
            const int j = i + rem;   // upr_tri(nq, i, j) == e for i <= j
            const int oc = 2 * d.nx + 2 * d.nu + d.np;
            for (int r = 0; r < d.no; ++r) v += L[o.wq + oc + r] * L[o.ob + d.no + r * d.nq + i] * L[o.ob + d.no + r * d.nq + j];
        }
        L[o.hob + e] = v;
    }
    if (k < d.N) {
        UPR_FOR(i, d.nu) {
            double g = h * P->Rdiag[i] * L[o.Uk + i] + L[o.sv + ou + i] - L[o.sv + ou + d.nu + i];
            if (d.np > 0 && i >= d.nq) {
                int fi = i - d.nq, ci = fi / 3, a = fi % 3;
                for (int r = 0; r < 5; ++r) {
                    double e[3];
                    upr_friction_row_jac(P, ci, r, e);
                    g += e[a] * L[o.sv + op + 5 * ci + r];
                }
            }
            L[o.gus + i] = g;
            L[o.Wu + i] = L[o.wq + ou + i] + L[o.wq + ou + d.nu + i];
        }
        // dynamics residual in absolute variables: A X_k + B U_k - X_{k+1}
        const double h2 = 0.5 * h * h, h3 = h * h * h / 6.0;
        UPR_FOR(j, d.nq) {
            double q = L[o.Xk + j], v = L[o.Xk + d.nq + j], a = L[o.Xk + 2 * d.nq + j], u = L[o.Uk + j];
            L[o.bk + j] = q + h * v + h2 * a + h3 * u - L[o.Xn + j];
            L[o.bk + d.nq + j] = v + h * a + h2 * u - L[o.Xn + d.nq + j];
            L[o.bk + 2 * d.nq + j] = a + h * u - L[o.Xn + 2 * d.nq + j];
        }
        // equality residual: g + C dx + Df du_f   (ee was loaded with g)
        UPR_FOR(r, d.ne) {
            double v = L[o.ee + r];
            for (int j = 0; j < d.nx; ++j) v += L[o.Cm + r * d.nx + j] * L[o.dxk + j];
            for (int j = 0; j < d.nfc; ++j) v += L[o.Df + r * d.nfc + j] * L[o.duk + d.nq + j];
            L[o.nuv + r] = v;  // keep the residual in nuv until the Schur step consumes it
        }
    }
    UPR_SYNC();
}

// Terminal knot: residual of [p_d - p; v; a] + CN dx_N (into misc area of hx: neN values stored in wv),
// and initialisation of (Pm, pv).  mat: also build Pm.
static inline UPR_HD void upr_qp_terminal(upr_qp_state& S, bool mat) {
    const upr_ctx& ctx = S.ctx; const upr_dims& d = S.d; double* L = S.L; const upr_qp_lds& o = S.o;
    const int N = d.N, nq = d.nq, nx = d.nx;
    upr_qp_load_stage(S, N);
    upr_qp_assemble(S, N);
    const double irho = 1.0 / UPR_QP_RHO_N;
    const double* yN = S.ws + d.ws_yN;
    // eN residual in wv: rows 0..2 position, then v (nq), a (nq).  J_p in hess (3 x nq), p_d - p in grad.
    if (d.neN > 0) {
        UPR_FOR(r, d.neN) {
            double v;
            if (r < 3) { v = L[o.grad + r]; for (int j = 0; j < nq; ++j) v -= L[o.hess + r * nq + j] * L[o.dxk + j]; }
            else v = L[o.Xk + nq + (r - 3)];
            L[o.wv + r] = v;
        }
        UPR_SYNC();
    }
    if (mat) {
        UPR_FOR(e, nx * nx) {
            int i = e / nx, j = e % nx;
            double v = (i == j) ? L[o.Wx + i] : 0.0;
            if (d.neN > 0) {
                if (i < nq && j < nq) { for (int r = 0; r < 3; ++r) v += irho * L[o.hess + r * nq + i] * L[o.hess + r * nq + j]; }
                else if (i == j) v += irho;
            }
            L[o.Pm + e] = v;
        }
    }
    UPR_FOR(i, nx) {
        double v = L[o.gxs + i];
        if (d.neN > 0) {
            if (i < nq) { for (int r = 0; r < 3; ++r) v -= L[o.hess + r * nq + i] * (yN[r] + irho * L[o.wv + r]); }
            else v += yN[3 + (i - nq)] + irho * L[o.wv + 3 + (i - nq)];
        }
        L[o.pv + i] = v;
    }
    UPR_SYNC();
}

// Backward sweep.  mat = true: factorise (predictor); false: vector pass only (corrector).
static inline UPR_HD void upr_qp_backward(upr_qp_state& S, bool mat) {
    const upr_ctx& ctx = S.ctx; const upr_dims& d = S.d; double* L = S.L; const upr_qp_lds& o = S.o;
    const upr_problem* P = S.P;
    const int N = d.N, nq = d.nq, nx = d.nx, ne = d.ne, nfc = d.nfc, nc = d.nc;
    const double h = P->dt, h2 = 0.5 * h * h, h3 = h * h * h / 6.0;
    upr_qp_terminal(S, mat);
    for (int k = N - 1; k >= 0; --k) {
        double* st = S.ws + d.ws_store + (size_t)k * d.ss_stride;
        upr_qp_load_stage(S, k);
        upr_qp_assemble(S, k);
        // wv = P+ b + p+
        if (mat) {
            UPR_FOR(i, nx) {
                double v = L[o.pv + i], pb = 0.0;
                for (int j = 0; j < nx; ++j) pb += L[o.Pm + i * nx + j] * L[o.bk + j];
                L[o.wv + i] = v + pb;
                L[o.hx + i] = pb;  // stash P+ b
            }
            UPR_SYNC();
            // keep P+ b for the corrector's vector pass
            UPR_FOR(i, nx) st[d.ss_pb + i] = L[o.hx + i];
            // Tm = B' P+ ; Hjj = Tm B (+ R + barrier) before the column ops
            UPR_FOR(e, nq * nx) {
                int j = e / nx, c = e % nx;
                L[o.Tm + e] = h3 * L[o.Pm + j * nx + c] + h2 * L[o.Pm + (nq + j) * nx + c] + h * L[o.Pm + (2 * nq + j) * nx + c];
            }
            UPR_SYNC();
            // phase A: Hjj = Tm B (+ R + barrier) from the untouched Tm; Pm a-columns (X <- X A)
            UPR_FOR(e, nq * nq) {
                int j = e / nq, m = e % nq;
                double v = h3 * L[o.Tm + j * nx + m] + h2 * L[o.Tm + j * nx + nq + m] + h * L[o.Tm + j * nx + 2 * nq + m];
                if (j == m) v += h * P->Rdiag[j] + L[o.Wu + j];
                L[o.Hjj + e] = v;
            }
            UPR_FOR(e, nx * nq) { int i = e / nq, j = e % nq; L[o.Pm + i * nx + 2 * nq + j] += h2 * L[o.Pm + i * nx + j] + h * L[o.Pm + i * nx + nq + j]; }
            UPR_SYNC();
            // phase B: Tm a-columns; Pm v-columns
            UPR_FOR(e, nq * nq) { int i = e / nq, j = e % nq; L[o.Tm + i * nx + 2 * nq + j] += h2 * L[o.Tm + i * nx + j] + h * L[o.Tm + i * nx + nq + j]; }
            UPR_FOR(e, nx * nq) { int i = e / nq, j = e % nq; L[o.Pm + i * nx + nq + j] += h * L[o.Pm + i * nx + j]; }
            UPR_SYNC();
            // phase C: Tm v-columns; Pm a-rows (X <- A' X)
            UPR_FOR(e, nq * nq) { int i = e / nq, j = e % nq; L[o.Tm + i * nx + nq + j] += h * L[o.Tm + i * nx + j]; }
            UPR_FOR(e, nq * nx) { int j = e / nx, c = e % nx; L[o.Pm + (2 * nq + j) * nx + c] += h2 * L[o.Pm + j * nx + c] + h * L[o.Pm + (nq + j) * nx + c]; }
            UPR_SYNC();
            // phase D: Pm v-rows
            UPR_FOR(e, nq * nx) { int j = e / nx, c = e % nx; L[o.Pm + (nq + j) * nx + c] += h * L[o.Pm + j * nx + c]; }
            UPR_SYNC();
            // stage cost + barrier on the state block
            UPR_FOR(e, nx * nx) {
                int i = e / nx, j = e % nx;
                double v = 0.0;
                if (i == j) v += h * P->Qdiag[i] + L[o.Wx + i];
                if (i < nq && j < nq) { v += h * L[o.hess + upr_tri(nq, i, j)]; if (d.no > 0) v += L[o.hob + upr_tri(nq, i, j)]; }
                L[o.Pm + e] += v;
            }
            UPR_SYNC();
            // Hjj = Lj Lj' ; Lji = Lj^-1 ; V = Lji Hux (Hux = Tm) ; Pm -= V'V
            upr_chol_inv(ctx, L + o.Hjj, L + o.Lji, nq, L + o.misc);
            UPR_FOR(e, nq * nx) {
                int i = e / nx, c = e % nx;
                double v = 0.0;
                for (int m = 0; m <= i; ++m) v += L[o.Lji + i * nq + m] * L[o.Tm + m * nx + c];
                L[o.Kx + e] = v;
            }
            UPR_SYNC();
            UPR_FOR(e, nx * nx) {
                int i = e / nx, j = e % nx;
                double v = 0.0;
                for (int m = 0; m < nq; ++m) v += L[o.Kx + m * nx + i] * L[o.Kx + m * nx + j];
                L[o.Pm + e] -= v;
            }
            // contact block Hff (3x3 per contact, or scalar per contact when nf == 1) -> Lfi = chol^-1
            if (d.nf == 3) {
                UPR_FOR(ci, nc) {
                    double* Hc = L + o.Hff + 9 * ci;
                    for (int a = 0; a < 9; ++a) Hc[a] = 0.0;
                    for (int a = 0; a < 3; ++a) Hc[4 * a] = h * P->Rdiag[nq + 3 * ci + a] + L[o.Wu + nq + 3 * ci + a];
                    for (int r = 0; r < 5; ++r) {
                        double e3[3];
                        upr_friction_row_jac(P, ci, r, e3);
                        double w = L[o.wq + 2 * nx + 2 * d.nu + 5 * ci + r];
                        for (int a = 0; a < 3; ++a) for (int b2 = 0; b2 < 3; ++b2) Hc[3 * a + b2] += w * e3[a] * e3[b2];
                    }
                    if (!upr_chol_inv3(Hc)) L[o.misc] = 1.0;
                }
            } else {
                UPR_FOR(ci, nc) L[o.Hff + ci] = 1.0 / sqrt(h * P->Rdiag[nq + ci] + L[o.Wu + nq + ci]);
            }
            UPR_SYNC();
            // Zf = Lfi Df'   (nfc x ne) ; S = Zf'Zf + rho_s I
            UPR_FOR(e, nfc * ne) {
                int i = e / ne, r = e % ne;
                double v;
                if (d.nf == 3) {
                    int ci = i / 3, a = i % 3;
                    const double* Bk = L + o.Hff + 9 * ci;
                    v = 0.0;
                    for (int b2 = 0; b2 <= a; ++b2) v += Bk[3 * a + b2] * L[o.Df + r * nfc + 3 * ci + b2];
                } else v = L[o.Hff + i] * L[o.Df + r * nfc + i];
                L[o.Yf + e] = v;
            }
            UPR_SYNC();
            UPR_FOR(e, ne * ne) {
                int r = e / ne, c = e % ne;
                double v = (r == c) ? upr_qp_rho_s(P, ne, nfc) : 0.0;
                for (int i = 0; i < nfc; ++i) v += L[o.Yf + i * ne + r] * L[o.Yf + i * ne + c];
                L[o.Sm + e] = v;
            }
            UPR_SYNC();
            upr_chol_inv(ctx, L + o.Sm, L + o.Lsi, ne, L + o.misc);
            // Vc = Lsi C ; Pm += Vc'Vc
            UPR_FOR(e, ne * nx) {
                int r = e / nx, c = e % nx;
                double v = 0.0;
                for (int m = 0; m <= r; ++m) v += L[o.Lsi + r * ne + m] * L[o.Cm + m * nx + c];
                L[o.SC + e] = v;
            }
            UPR_SYNC();
            UPR_FOR(e, nx * nx) {
                int i = e / nx, j = e % nx;
                double v = 0.0;
                for (int r = 0; r < ne; ++r) v += L[o.SC + r * nx + i] * L[o.SC + r * nx + j];
                L[o.Pm + e] += v;
            }
            UPR_SYNC();
            // keep the cost-to-go symmetric: rounding asymmetry is amplified by the recursion otherwise
            UPR_FOR(e, nx * nx) {
                int i = e / nx, j = e % nx;
                if (i < j) { double v = 0.5 * (L[o.Pm + i * nx + j] + L[o.Pm + j * nx + i]); L[o.Pm + i * nx + j] = v; L[o.Pm + j * nx + i] = v; }
            }
            // store the factors
            UPR_FOR(e, nq * nx) st[d.ss_kx + e] = L[o.Kx + e];
            UPR_FOR(e, nq * nq) st[d.ss_hjj + e] = L[o.Lji + e];
            UPR_FOR(e, (d.nf == 3 ? 9 * nc : nc)) st[d.ss_hff + e] = L[o.Hff + e];
            UPR_FOR(e, ne * ne) st[d.ss_sinv + e] = L[o.Lsi + e];
            UPR_SYNC();
        } else {
            UPR_FOR(i, nx) L[o.wv + i] = L[o.pv + i] + st[d.ss_pb + i];
            UPR_FOR(e, nq * nx) L[o.Kx + e] = st[d.ss_kx + e];
            UPR_FOR(e, nq * nq) L[o.Lji + e] = st[d.ss_hjj + e];
            UPR_FOR(e, (d.nf == 3 ? 9 * nc : nc)) L[o.Hff + e] = st[d.ss_hff + e];
            UPR_FOR(e, ne * ne) L[o.Lsi + e] = st[d.ss_sinv + e];
            UPR_SYNC();
        }
        // ---- vector part (both passes) ----
        upr_At_vec(ctx, nq, h, L + o.wv, L + o.hx);
        UPR_FOR(j, nq) L[o.huj + j] = L[o.gus + j] + upr_Bt_vec_j(nq, h, L + o.wv, j);
        UPR_FOR(i, nfc) L[o.huf + i] = L[o.gus + nq + i];
        UPR_SYNC();
        UPR_FOR(i, nx) L[o.hx + i] += L[o.gxs + i];
        // yj = Lji huj ; yf = Lfi huf
        UPR_FOR(j, nq) {
            double v = 0.0;
            for (int m = 0; m <= j; ++m) v += L[o.Lji + j * nq + m] * L[o.huj + m];
            L[o.ku0 + j] = v;
        }
        UPR_FOR(i, nfc) L[o.uf0 + i] = upr_blk_lo(d, L + o.Hff, L + o.huf, i);
        UPR_SYNC();
        // huf <- Lfi' yf  (= Hff^-1 huf) ; ee = e_res - Df (Hff^-1 huf)
        UPR_FOR(i, nfc) L[o.huf + i] = upr_blk_up(d, L + o.Hff, L + o.uf0, i);
        UPR_SYNC();
        UPR_FOR(r, ne) {
            double v = L[o.nuv + r] + upr_qp_rho_prox(P, ne, nfc) * S.ws[d.ws_nu + k * ne + r];
            for (int i = 0; i < nfc; ++i) v -= L[o.Df + r * nfc + i] * L[o.huf + i];
            L[o.ee + r] = v;
        }
        UPR_SYNC();
        // ys = Lsi ee ; then ee <- Lsi' ys (= S^-1 ee)
        UPR_FOR(r, ne) {
            double v = 0.0;
            for (int m = 0; m <= r; ++m) v += L[o.Lsi + r * ne + m] * L[o.ee + m];
            L[o.snu + r] = v;
        }
        UPR_SYNC();
        UPR_FOR(r, ne) {
            double v = 0.0;
            for (int m = r; m < ne; ++m) v += L[o.Lsi + m * ne + r] * L[o.snu + m];
            L[o.ee + r] = v;
        }
        UPR_SYNC();
        // pv = hx - V' yj + C' (S^-1 ee)
        UPR_FOR(i, nx) {
            double v = L[o.hx + i];
            for (int m = 0; m < nq; ++m) v -= L[o.Kx + m * nx + i] * L[o.ku0 + m];
            for (int r = 0; r < ne; ++r) v += L[o.Cm + r * nx + i] * L[o.ee + r];
            L[o.pv + i] = v;
        }
        UPR_FOR(j, nq) st[d.ss_ku0 + j] = L[o.ku0 + j];
        UPR_FOR(i, nfc) st[d.ss_uf0 + i] = L[o.uf0 + i];
        UPR_FOR(r, ne) st[d.ss_snu + r] = L[o.snu + r];
        UPR_SYNC();
    }
}

// Forward sweep: writes the step (sx, su), the full-step stage multipliers nu+ (`nu_new`, [N][ne])
// and the terminal multiplier step `dyN` ([neN]).
static inline UPR_HD void upr_qp_forward(upr_qp_state& S, double* nu_new, double* dyN) {
    const upr_ctx& ctx = S.ctx; const upr_dims& d = S.d; double* L = S.L; const upr_qp_lds& o = S.o;
    const upr_problem* P = S.P;
    const int N = d.N, nq = d.nq, nx = d.nx, ne = d.ne, nfc = d.nfc;
    const double h = P->dt, h2 = 0.5 * h * h, h3 = h * h * h / 6.0;
    double* sx = S.ws + d.ws_sx; double* su = S.ws + d.ws_su;
    UPR_FOR(i, nx) { L[o.sxk + i] = 0.0; sx[i] = 0.0; }  // x_0 is fixed
    UPR_SYNC();
    for (int k = 0; k < N; ++k) {
        const double* st = S.ws + d.ws_store + (size_t)k * d.ss_stride;
        const double* rec = S.lin + (size_t)k * d.lin_stride;
        UPR_FOR(e, nq * nx) L[o.Kx + e] = st[d.ss_kx + e];
        UPR_FOR(e, ne * nx) L[o.Cm + e] = rec[d.lin_gx + e];
        UPR_FOR(e, ne * ne) L[o.Lsi + e] = st[d.ss_sinv + e];
        UPR_FOR(e, nq * nq) L[o.Lji + e] = st[d.ss_hjj + e];
        UPR_FOR(e, (d.nf == 3 ? 9 * d.nc : d.nc)) L[o.Hff + e] = st[d.ss_hff + e];
        UPR_FOR(j, nq) L[o.ku0 + j] = st[d.ss_ku0 + j];
        UPR_FOR(i, nfc) L[o.uf0 + i] = st[d.ss_uf0 + i];
        UPR_FOR(r, ne) L[o.snu + r] = st[d.ss_snu + r];
        // dynamics residual needs the current iterate (absolute)
        {
            const double* dx = S.ws + d.ws_dx; const double* du = S.ws + d.ws_du;
            UPR_FOR(i, nx) { L[o.Xk + i] = S.xs[k * nx + i] + dx[k * nx + i]; L[o.Xn + i] = S.xs[(k + 1) * nx + i] + dx[(k + 1) * nx + i]; }
            UPR_FOR(i, d.nu) L[o.Uk + i] = S.us[k * d.nu + i] + du[k * d.nu + i];
        }
        UPR_SYNC();
        // nu = Lsi' (Lsi (C sx) + ys)
        UPR_FOR(r, ne) {
            double v = 0.0;
            for (int j = 0; j < nx; ++j) v += L[o.Cm + r * nx + j] * L[o.sxk + j];
            L[o.ee + r] = v;
        }
        // tj = V sx + yj
        UPR_FOR(j, nq) {
            double v = L[o.ku0 + j];
            for (int c = 0; c < nx; ++c) v += L[o.Kx + j * nx + c] * L[o.sxk + c];
            L[o.huj + j] = v;
        }
        UPR_SYNC();
        UPR_FOR(r, ne) {
            double v = L[o.snu + r];
            for (int m = 0; m <= r; ++m) v += L[o.Lsi + r * ne + m] * L[o.ee + m];
            L[o.gjr + r] = v;   // (ne entries: wv holds nx only)
        }
        // su_j = -Lji' tj
        UPR_FOR(j, nq) {
            double v = 0.0;
            for (int m = j; m < nq; ++m) v += L[o.Lji + m * nq + j] * L[o.huj + m];
            L[o.suk + j] = -v;
        }
        UPR_SYNC();
        UPR_FOR(r, ne) {
            double v = 0.0;
            for (int m = r; m < ne; ++m) v += L[o.Lsi + m * ne + r] * L[o.gjr + m];
            L[o.nuv + r] = v;
            nu_new[k * ne + r] = v;
        }
        UPR_SYNC();
        // su_f = -Lfi' (yf + Lfi Df' nu)
        UPR_FOR(i, nfc) {
            double v = 0.0;
            for (int r = 0; r < ne; ++r) v += L[o.Df + r * nfc + i] * L[o.nuv + r];
            L[o.huf + i] = v;
        }
        UPR_SYNC();
        UPR_FOR(i, nfc) L[o.gus + i] = L[o.uf0 + i] + upr_blk_lo(d, L + o.Hff, L + o.huf, i);
        UPR_SYNC();
        UPR_FOR(i, nfc) L[o.suk + nq + i] = -upr_blk_up(d, L + o.Hff, L + o.gus, i);
        UPR_SYNC();
        UPR_FOR(i, d.nu) su[k * d.nu + i] = L[o.suk + i];
        // sx+ = A sx + B su_j + b_k
        UPR_FOR(j, nq) {
            double q = L[o.sxk + j], v = L[o.sxk + nq + j], a = L[o.sxk + 2 * nq + j], u = L[o.suk + j];
            double X = L[o.Xk + j], V = L[o.Xk + nq + j], Ac = L[o.Xk + 2 * nq + j], U = L[o.Uk + j];
            L[o.hx + j] = q + h * v + h2 * a + h3 * u + (X + h * V + h2 * Ac + h3 * U - L[o.Xn + j]);
            L[o.hx + nq + j] = v + h * a + h2 * u + (V + h * Ac + h2 * U - L[o.Xn + nq + j]);
            L[o.hx + 2 * nq + j] = a + h * u + (Ac + h * U - L[o.Xn + 2 * nq + j]);
        }
        UPR_SYNC();
        UPR_FOR(i, nx) { double v = L[o.hx + i]; L[o.sxk + i] = v; sx[(k + 1) * nx + i] = v; }
        UPR_SYNC();
    }
    // terminal multiplier step: dyN = (CN sx_N + eN_res) / rhoN
    if (d.neN > 0) {
        const double* rec = S.lin + (size_t)N * d.lin_stride;
        const double* dx = S.ws + d.ws_dx;
        UPR_FOR(r, d.neN) {
            double v;
            if (r < 3) {
                v = rec[d.lin_grad + r];
                for (int j = 0; j < nq; ++j) v -= rec[d.lin_hess + r * nq + j] * (dx[N * nx + j] + L[o.sxk + j]);
            } else v = S.xs[N * nx + nq + (r - 3)] + dx[N * nx + nq + (r - 3)] + L[o.sxk + nq + (r - 3)];
            dyN[r] = v / UPR_QP_RHO_N;
        }
    }
    UPR_SYNC();
}

// Costate sweep for the full step: pi+_k = gxs_k + Htilde_xx sx_k + A' pi+_{k+1} + C_k' nu+_k  (k = N..1)
static inline UPR_HD void upr_qp_costates(upr_qp_state& S, const double* nu_new, const double* dyN, double* pi_new) {
    const upr_ctx& ctx = S.ctx; const upr_dims& d = S.d; double* L = S.L; const upr_qp_lds& o = S.o;
    const upr_problem* P = S.P;
    const int N = d.N, nq = d.nq, nx = d.nx, ne = d.ne;
    const double h = P->dt;
    const double* yN = S.ws + d.ws_yN;
    upr_qp_load_stage(S, N);
    upr_qp_assemble(S, N);
    UPR_FOR(i, nx) {
        double v = L[o.gxs + i] + L[o.Wx + i] * L[o.sxk + i];
        if (d.neN > 0) {
            if (i < nq) { for (int r = 0; r < 3; ++r) v -= L[o.hess + r * nq + i] * (yN[r] + dyN[r]); }
            else v += yN[3 + (i - nq)] + dyN[3 + (i - nq)];
        }
        L[o.pv + i] = v;
        pi_new[N * nx + i] = v;
    }
    UPR_SYNC();
    for (int k = N - 1; k >= 1; --k) {
        upr_qp_load_stage(S, k);
        upr_qp_assemble(S, k);
        upr_At_vec(ctx, nq, h, L + o.pv, L + o.hx);
        UPR_FOR(r, ne) L[o.snu + r] = nu_new[k * ne + r];
        UPR_SYNC();
        UPR_FOR(i, nx) {
            double v = L[o.hx + i] + L[o.gxs + i] + (h * P->Qdiag[i] + L[o.Wx + i]) * L[o.sxk + i];
            if (i < nq) for (int j = 0; j < nq; ++j) v += (h * L[o.hess + upr_tri(nq, i, j)] + (d.no > 0 ? L[o.hob + upr_tri(nq, i, j)] : 0.0)) * L[o.sxk + j];
            for (int r = 0; r < ne; ++r) v += L[o.Cm + r * nx + i] * L[o.snu + r];
            L[o.wv + i] = v;
        }
        UPR_SYNC();
        UPR_FOR(i, nx) { L[o.pv + i] = L[o.wv + i]; pi_new[k * nx + i] = L[o.wv + i]; }
        UPR_SYNC();
    }
}

// One sweep over every inequality with the current step (sx, su).  The complementarity target rc is
// lam*t in predictor mode (S.mode == 0) and the stored corrector target otherwise.
//   what = 0: largest feasible step (per-thread partial of alpha_max)
//   what = 1: partial sum of (lam + a dlam)(t + a dt) for a = alpha
//   what = 2: apply t += a dt, lam += a dlam
//   what = 5: partial MIN of the trial products (lam + a dlam)(t + a dt) (and of the slack pairs'); aux[0] accumulates their sum
//   what = 3: partial max of |c - t| (r_ineq); aux[0] accumulates the partial sum of lam*t (+ gam*tau of softened rows),
//             aux[1] the partial max of the slack stationarity |Z sigma + z - lam - gam|
static inline UPR_HD double upr_qp_ineq_sweep(upr_qp_state& S, int what, double alpha, double* aux) {
    const upr_ctx& ctx = S.ctx; const upr_dims& d = S.d; double* L = S.L; const upr_qp_lds& o = S.o;
    const upr_problem* P = S.P;
    const double* dx = S.ws + d.ws_dx; const double* du = S.ws + d.ws_du;
    const double* sx = S.ws + d.ws_sx; const double* su = S.ws + d.ws_su;
    double acc = (what == 0) ? 1e30 : (what == 5 ? 1e300 : 0.0);
    for (int k = 0; k <= d.N; ++k) {
        UPR_FOR(i, d.nx) { L[o.Xk + i] = S.xs[k * d.nx + i] + dx[k * d.nx + i]; L[o.sxk + i] = sx[k * d.nx + i]; }
        if (k < d.N) UPR_FOR(i, d.nu) { L[o.Uk + i] = S.us[k * d.nu + i] + du[k * d.nu + i]; L[o.suk + i] = su[k * d.nu + i]; }
        if (d.no > 0 && k >= 1 && k < d.N) {
            const double* rec = S.lin + (size_t)k * d.lin_stride;
            UPR_FOR(i, d.no * (1 + d.nq)) L[o.ob + i] = rec[d.lin_obs + i];
            UPR_FOR(i, d.nq) L[o.dxk + i] = dx[k * d.nx + i];
        }
        UPR_SYNC();
        double* t = S.ws + d.ws_t + (size_t)k * d.ni_stage;
        double* lam = S.ws + d.ws_lam + (size_t)k * d.ni_stage;
        const double* rcg = S.ws + d.ws_rc + (size_t)k * d.ni_stage;
        UPR_FOR(j, d.ni_stage) {
            if (!upr_ineq_active(d, k, j)) continue;
            double tj = t[j], lj = lam[j];
            double c = upr_ineq_value(P, d, j, L + o.Xk, L + o.Uk, L + o.ob, L + o.dxk);
            double rp = c - tj;
            if (d.soft && upr_slot_soft(P, d, j)) {
                double* sg = S.ws + d.ws_sig + (size_t)k * d.ni_stage; double* ta = S.ws + d.ws_tau + (size_t)k * d.ni_stage;
                double* ga = S.ws + d.ws_gam + (size_t)k * d.ni_stage; const double* rcsg = S.ws + d.ws_rcs + (size_t)k * d.ni_stage;
                const double sig = sg[j], tau = ta[j], gam = ga[j];
                rp += sig;
                const double rc = (S.mode == 0) ? lj * tj : rcg[j], rcs = (S.mode == 0) ? gam * tau : rcsg[j];
                const upr_soft_row r = upr_soft_terms(P, d, j, tj, lj, sig, tau, gam, rp, rc, rcs);
                if (what == 3) {
                    double a = fmax(fabs(rp), fabs(r.rps)); if (a > acc) acc = a;
                    aux[0] += lj * tj + gam * tau;
                    aux[1] = fmax(aux[1], fabs(r.Z * sig + r.z - lj - gam));
                    continue;
                }
                const double gdz = upr_ineq_dir(P, d, j, L + o.sxk, L + o.suk, L + o.ob);
                const double ds = -(r.a + r.w * gdz) / r.D;
                const double dt = gdz + rp + ds, dtau = ds + r.rps;
                const double dl = -(rc + lj * dt) / tj, dg = -(rcs + gam * dtau) / tau;
                if (what == 0) {
                    if (dt < 0.0) { double a = -tj / dt; if (a < acc) acc = a; }
                    if (dl < 0.0) { double a = -lj / dl; if (a < acc) acc = a; }
                    if (dtau < 0.0) { double a = -tau / dtau; if (a < acc) acc = a; }
                    if (dg < 0.0) { double a = -gam / dg; if (a < acc) acc = a; }
                } else if (what == 1) {
                    acc += (lj + alpha * dl) * (tj + alpha * dt) + (gam + alpha * dg) * (tau + alpha * dtau);
                } else if (what == 5) {
                    const double v = (lj + alpha * dl) * (tj + alpha * dt), vs = (gam + alpha * dg) * (tau + alpha * dtau);
                    acc = fmin(acc, fmin(v, vs)); aux[0] += v + vs;
                } else {
                    t[j] = tj + alpha * dt; lam[j] = lj + alpha * dl;
                    sg[j] = sig + alpha * ds; ta[j] = tau + alpha * dtau; ga[j] = gam + alpha * dg;
                }
                continue;
            }
            if (what == 3) { double a = fabs(rp); if (a > acc) acc = a; *aux += lj * tj; continue; }
            double dt = upr_ineq_dir(P, d, j, L + o.sxk, L + o.suk, L + o.ob) + rp;
            double rc = (S.mode == 0) ? lj * tj : rcg[j];
            double dl = -(rc + lj * dt) / tj;
            if (what == 0) {
                if (dt < 0.0) { double a = -tj / dt; if (a < acc) acc = a; }
                if (dl < 0.0) { double a = -lj / dl; if (a < acc) acc = a; }
            } else if (what == 1) {
                acc += (lj + alpha * dl) * (tj + alpha * dt);
            } else if (what == 5) {
                const double v = (lj + alpha * dl) * (tj + alpha * dt);
                acc = fmin(acc, v); aux[0] += v;
            } else {
                t[j] = tj + alpha * dt; lam[j] = lj + alpha * dl;
            }
        }
        UPR_SYNC();
    }
    return acc;
}

// Explicit KKT residuals at the current iterate with the current multipliers:
// res = [stationarity, equality (dynamics, stage, terminal), inequality, mu]
static inline UPR_HD void upr_qp_residuals(upr_qp_state& S, int ntot, double* res) {
    const upr_ctx& ctx = S.ctx; const upr_dims& d = S.d; double* L = S.L; const upr_qp_lds& o = S.o;
    const upr_problem* P = S.P;
    const int N = d.N, nq = d.nq, nx = d.nx, ne = d.ne, nfc = d.nfc;
    const double h = P->dt;
    const double* pi = S.ws + d.ws_pi; const double* nu = S.ws + d.ws_nu; const double* yN = S.ws + d.ws_yN;
    double r_stat = 0.0, r_eq = 0.0;
    const int save_mode = S.mode;
    S.mode = 2;
    for (int k = N; k >= 0; --k) {
        upr_qp_load_stage(S, k);
        upr_qp_assemble(S, k);
        if (k < N) {
            UPR_FOR(i, nx) L[o.wv + i] = pi[(k + 1) * nx + i];
            UPR_FOR(r, ne) L[o.snu + r] = nu[k * ne + r];
            UPR_SYNC();
            upr_At_vec(ctx, nq, h, L + o.wv, L + o.hx);
            UPR_SYNC();
            if (k >= 1) UPR_FOR(i, nx) {
                double v = L[o.gxs + i] + L[o.hx + i] - pi[k * nx + i];
                for (int r = 0; r < ne; ++r) v += L[o.Cm + r * nx + i] * L[o.snu + r];
                r_stat = fmax(r_stat, fabs(v));
            }
            UPR_FOR(i, d.nu) {
                double v = L[o.gus + i];
                if (i < nq) v += upr_Bt_vec_j(nq, h, L + o.wv, i);
                else for (int r = 0; r < ne; ++r) v += L[o.Df + r * nfc + (i - nq)] * L[o.snu + r];
                r_stat = fmax(r_stat, fabs(v));
            }
            UPR_FOR(i, nx) r_eq = fmax(r_eq, fabs(L[o.bk + i]));
            UPR_FOR(r, ne) r_eq = fmax(r_eq, fabs(L[o.nuv + r] - upr_qp_rho_soft(P) * L[o.snu + r]));
        } else {
            if (d.neN > 0) {
                UPR_FOR(r, d.neN) {
                    double v;
                    if (r < 3) { v = L[o.grad + r]; for (int j = 0; j < nq; ++j) v -= L[o.hess + r * nq + j] * L[o.dxk + j]; }
                    else v = L[o.Xk + nq + (r - 3)];
                    r_eq = fmax(r_eq, fabs(v));
                }
            }
            UPR_FOR(i, nx) {
                double v = L[o.gxs + i] - pi[N * nx + i];
                if (d.neN > 0) {
                    if (i < nq) { for (int r = 0; r < 3; ++r) v -= L[o.hess + r * nq + i] * yN[r]; }
                    else v += yN[3 + (i - nq)];
                }
                r_stat = fmax(r_stat, fabs(v));
            }
        }
        UPR_SYNC();
    }
    S.mode = save_mode;
    double lt[2] = {0.0, 0.0};
    double r_in = upr_qp_ineq_sweep(S, 3, 0.0, lt);
    r_stat = fmax(r_stat, lt[1]);
    res[0] = upr_reduce(ctx, L + o.red, r_stat, 1);
    res[1] = upr_reduce(ctx, L + o.red, r_eq, 1);
    res[2] = upr_reduce(ctx, L + o.red, r_in, 1);
    res[3] = upr_reduce(ctx, L + o.red, lt[0], 0) / (ntot > 0 ? ntot : 1);
}

// The whole QP for instance b.  L: workgroup scratch of upr_qp_lds_layout(...).total doubles.
static inline UPR_HD void upr_qp_solve(const upr_ctx& ctx, const upr_qp_args& A, int b, double* L) {
    upr_qp_state S;
    S.ctx = ctx; S.P = A.P; S.d = A.d; S.o = upr_qp_lds_layout(A.d, ctx.nt); S.L = L;
    const upr_dims& d = S.d; const upr_qp_lds& o = S.o; const upr_problem* P = A.P;
    const int N = d.N, nx = d.nx, nu = d.nu, ne = d.ne;
    S.xs = A.xs + (size_t)b * (N + 1) * nx; S.us = A.us + (size_t)b * N * nu; S.x0 = A.x0 + (size_t)b * nx;
    S.lin = A.lin + (size_t)b * (N + 1) * d.lin_stride; S.Dfg = A.Df + (size_t)b * ne * d.nfc;
    S.ws = A.ws + (size_t)b * d.ws_stride;
    S.mode = 0; S.sigma_mu = 0.0;
    double* ws = S.ws;
    // ---- initial point: dz = 0 (dx_0 = x0 - xs_0), multipliers 0, t = max(c, thr), lam = mu0 / t
    UPR_FOR(i, d.ws_store) ws[i] = 0.0;
    UPR_FOR(i, ne * d.nfc) L[o.Df + i] = S.Dfg[i];
    if (ctx.tid == 0) L[o.misc] = 0.0;
    UPR_SYNC();
    UPR_FOR(i, nx) ws[d.ws_dx + i] = S.x0[i] - S.xs[i];
    UPR_SYNC();
    for (int k = 0; k <= N; ++k) {
        UPR_FOR(i, nx) L[o.Xk + i] = S.xs[k * nx + i] + ws[d.ws_dx + k * nx + i];
        if (k < N) UPR_FOR(i, nu) L[o.Uk + i] = S.us[k * nu + i];
        if (d.no > 0 && k >= 1 && k < N) {
            const double* rec = S.lin + (size_t)k * d.lin_stride;
            UPR_FOR(i, d.no * (1 + d.nq)) L[o.ob + i] = rec[d.lin_obs + i];
            UPR_FOR(i, d.nq) L[o.dxk + i] = ws[d.ws_dx + k * nx + i];
        }
        UPR_SYNC();
        UPR_FOR(j, d.ni_stage) {
            double t = 1.0, lam = 0.0;
            if (upr_ineq_active(d, k, j)) {
                double c = upr_ineq_value(P, d, j, L + o.Xk, L + o.Uk, L + o.ob, L + o.dxk);
                t = c > UPR_QP_THR ? c : UPR_QP_THR;
                lam = UPR_QP_MU0 / t;
            }
            ws[d.ws_t + k * d.ni_stage + j] = t; ws[d.ws_lam + k * d.ni_stage + j] = lam;
            if (d.soft) {                                      // slack 0, its own barrier pair at (thr, mu0 / thr)
                const bool sf = upr_ineq_active(d, k, j) && upr_slot_soft(P, d, j);
                ws[d.ws_tau + k * d.ni_stage + j] = sf ? UPR_QP_THR : 1.0; ws[d.ws_gam + k * d.ni_stage + j] = sf ? UPR_QP_MU0 / UPR_QP_THR : 0.0;
            }
        }
        UPR_SYNC();
    }
    int ntot = N * (2 * nu + d.np) + N * 2 * nx + (N - 1) * d.no;
    if (P->soft_state_box) ntot += N * 2 * nx;               // each softened row adds the pair (tau, gam)
    if (P->soft_input_box) ntot += N * 2 * nu;
    if (P->soft_poly) ntot += N * d.np + (N - 1) * d.no;
    double* pi_new = ws + d.ws_pin; double* nu_new = ws + d.ws_nun; double* dyN = ws + d.ws_dyN;
    double res[4] = {0, 0, 0, 0};
    int it = 0, status = 1;
    const double tol = P->qp_tol;
    const double tol_stat = P->qp_tol_stat > 0.0 ? P->qp_tol_stat : tol;   // HPIPM tol_stat (upright_mi.h)
    for (;; ++it) {
        upr_qp_residuals(S, ntot, res);
#ifdef UPR_HOST_EMU
        if (getenv("UPR_EMU_DEBUG")) printf("it %d res %.3e %.3e %.3e %.3e sigma_mu %.3e\n", it, res[0], res[1], res[2], res[3], S.sigma_mu);
#endif
        if (it > 0 && res[0] < tol_stat && res[1] < tol && res[2] < tol && res[3] < tol) { status = 0; break; }
        if (it >= P->qp_iter_max) break;
        const double mu = res[3];
        // ---- predictor
        S.mode = 0;
        upr_qp_backward(S, true);
        if (L[o.misc] != 0.0) { status = 2; break; }
        upr_qp_forward(S, nu_new, dyN);
#ifdef UPR_HOST_EMU
        if (getenv("UPR_EMU_STOP_AFF")) return;
#endif
        double a_aff = upr_reduce(ctx, L + o.red, upr_qp_ineq_sweep(S, 0, 0.0, nullptr), 2);
        if (a_aff > 1.0) a_aff = 1.0;
        double mu_aff = upr_reduce(ctx, L + o.red, upr_qp_ineq_sweep(S, 1, a_aff, nullptr), 0) / ntot;
        double sg = mu_aff / mu;
        S.sigma_mu = sg * sg * sg * mu;
        if (S.sigma_mu < UPR_QP_SIGMA_FLOOR * tol) S.sigma_mu = UPR_QP_SIGMA_FLOOR * tol;
        // ---- corrector
        S.mode = 1;
        upr_qp_backward(S, false);
        upr_qp_forward(S, nu_new, dyN);
        S.mode = 3;
        upr_qp_costates(S, nu_new, dyN, pi_new);
        // step length: the largest feasible one, capped at 1, then shortened by 0.995 (HPIPM's update_var rule:
        // never a full step, so slacks and multipliers stay strictly positive)
        double a = upr_reduce(ctx, L + o.red, upr_qp_ineq_sweep(S, 0, 0.0, nullptr), 2);
        if (a > 1.0) a = 1.0;
        a *= 0.995;
        if (UPR_QP_NGAM > 0.0 && it < UPR_QP_NIT) {   // centrality safeguard (see UPR_QP_NGAM)
            double sm = 0.0;
            const double mn = upr_reduce(ctx, L + o.red, upr_qp_ineq_sweep(S, 5, a, &sm), 2);
            sm = upr_reduce(ctx, L + o.red, sm, 0);
            if (!(mn >= UPR_QP_NGAM * (sm / (ntot > 0 ? ntot : 1)))) a *= UPR_QP_NBT;
        }
        upr_qp_ineq_sweep(S, 2, a, nullptr);
        UPR_FOR(i, (N + 1) * nx) {
            ws[d.ws_dx + i] += a * ws[d.ws_sx + i];
            ws[d.ws_pi + i] += a * (pi_new[i] - ws[d.ws_pi + i]);
        }
        UPR_FOR(i, N * nu) ws[d.ws_du + i] += a * ws[d.ws_su + i];
        UPR_FOR(i, N * ne) ws[d.ws_nu + i] += a * (nu_new[i] - ws[d.ws_nu + i]);
        UPR_FOR(i, d.neN) ws[d.ws_yN + i] += a * dyN[i];
        UPR_SYNC();
    }
    if (ctx.tid == 0) {
        double* st = A.stats + (size_t)b * UPR_NSTATS;
        st[1] = it; st[2] = status; st[6] = res[0]; st[7] = res[1]; st[8] = res[2]; st[9] = res[3];
        upr_qp_store_key(A, b, it);
    }
    UPR_SYNC();
}

#ifndef UPR_HOST_EMU
template <int NT>
__global__ void __launch_bounds__(NT) upr_qp_kernel(upr_qp_args A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    upr_ctx ctx; ctx.tid = threadIdx.x; ctx.nt = NT;
    upr_qp_solve(ctx, A, upr_qp_instance(A, blockIdx.x), smem);
}
#endif
