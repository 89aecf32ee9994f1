// upr_qp2.h -- production QP kernel: same algorithm as upr_qp.h (the generic, runtime-dimension
// kernel), restructured around what is sequential and what is not.
//
//   * dimensions are template parameters: every offset is a literal, loops unroll, no SGPR spills;
//   * the iterate Z = (x + dx, u + du) and the step S live in LDS for all knots, so every pass over the
//     inequalities (residuals, step length, mu, update) is ONE flat workgroup-wide loop over
//     (knot, row) instead of a knot-by-knot sweep;
//   * everything that does not depend on the cost-to-go is hoisted out of the Riccati recursion into
//     flat stage-parallel phases: barrier weights and targets, reduced gradients, the 3x3 contact
//     blocks and their Cholesky inverses, the Schur complement S_k = Df Hff_k^-1 Df' of the
//     object-dynamics equality with its factor, Vc_k = Ls_k^-1 C_k, and the force / multiplier part of
//     the back-substitution;
//   * what remains sequential per knot is the jerk/state recursion: P+ b, B'P+, A'P+A (block-scalar
//     combinations), one nq x nq Cholesky (one lane, registers), V = Lj^-1 Hux, and the symmetric update
//     P = A'P+A + Q~ - V'V + Vc'Vc.
//
// Reference interfaces replaced: see upr_qp.h.
#pragma once
#include "upr_kin.h"
#include "upr_qp.h"

template <int NQ_, int NB_, int NC_, int NF_>
struct upr_qp2_dims {
    static constexpr int NQ = NQ_, NB = NB_, NC = NC_, NF = NF_;
    static constexpr int NX = 3 * NQ, NFC = NF * NC, NU = NQ + NFC, NE = 6 * NB;
    static constexpr int NP = (NF == 3) ? 5 * NC : 0;
    static constexpr int NI = 2 * NX + 2 * NU + NP;
    static constexpr int NLF = (NF == 3) ? 9 * NC : NC;  // contact-block factor storage
    static constexpr int NH = NQ * (NQ + 1) / 2;
    // per-knot "pre" record (global): everything the sequential sweeps consume
    static constexpr int PR_GXS = 0, PR_GUS = PR_GXS + NX, PR_WX = PR_GUS + NU, PR_WUJ = PR_WX + NX, PR_BK = PR_WUJ + NQ,
                         PR_EK = PR_BK + NX, PR_LFI = PR_EK + NE, PR_LSI = PR_LFI + NLF, PR_VC = PR_LSI + NE * NE,
                         PR_YF = PR_VC + NE * NX, PR_YS = PR_YF + NFC, PR_CS = PR_YS + NE, PR_STRIDE = ((PR_CS + NX + 1) & ~1);
    // per-knot Riccati store (global)
    static constexpr int SS_V = 0, SS_LJI = SS_V + NQ * NX, SS_YJ = SS_LJI + NQ * NQ, SS_PB = SS_YJ + NQ, SS_STRIDE = ((SS_PB + NX + 1) & ~1);
};

// global workspace of one instance (doubles), N runtime
template <class D>
struct upr_qp2_ws {
    int N, neN;
    int dx, du, t, lam, rc, wq, sv, pi, nu, yN, pin, nun, dyN, pre, store, total;
    UPR_HD upr_qp2_ws(int N_, int neN_) : N(N_), neN(neN_) {
        int n1 = N + 1, o = 0;
        auto take = [&](int n) { int r = o; o += (n + 1) & ~1; return r; };
        dx = 0; du = n1 * D::NX; o = (n1 * D::NX + N * D::NU + 1) & ~1;  // same place as the generic kernel (line search reads them)
        t = take(n1 * D::NI); lam = take(n1 * D::NI); rc = take(n1 * D::NI); wq = take(n1 * D::NI); sv = take(n1 * D::NI);
        pi = take(n1 * D::NX); nu = take(N * D::NE); yN = take(neN); pin = take(n1 * D::NX); nun = take(N * D::NE); dyN = take(neN);
        pre = take(n1 * D::PR_STRIDE); store = take(N * D::SS_STRIDE);
        total = (o + 15) & ~15;
    }
};

// LDS layout of one workgroup (doubles)
template <class D>
struct upr_qp2_lds {
    int Z, S, Pm, Tm, V, Hjj, Lji, Vc, pv, wv, hx, huj, yj, bk, cs, gxs, gus, wx, wuj, red, misc, total;
    UPR_HD upr_qp2_lds(int N, int nt) {
        int n1 = N + 1, o = 0;
        auto take = [&](int n) { int r = o; o += (n + 1) & ~1; return r; };
        Z = take(n1 * D::NX + N * D::NU); S = take(n1 * D::NX + N * D::NU);
        Pm = take(D::NX * D::NX); Tm = take(D::NQ * D::NX); V = take(D::NQ * D::NX); Hjj = take(D::NQ * D::NQ); Lji = take(D::NQ * D::NQ);
        Vc = take(D::NE * D::NX);
        pv = take(D::NX); wv = take(D::NX); hx = take(D::NX); huj = take(D::NQ); yj = take(D::NQ); bk = take(D::NX); cs = take(D::NX);
        gxs = take(D::NX); gus = take(D::NU); wx = take(D::NX); wuj = take(D::NQ);
        red = take(4 * nt); misc = take(16);
        total = o;
    }
};

// ---- serial (one lane, registers) Cholesky + triangular inverse of an SPD n x n matrix ------------------
template <int n>
static inline UPR_HD bool upr_chol_inv_serial(const double* M, double* Li) {
    double a[n][n];
#pragma unroll
    for (int i = 0; i < n; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) a[i][j] = M[i * n + j];
    bool ok = true;
#pragma unroll
    for (int p = 0; p < n; ++p) {
        double s = a[p][p];
#pragma unroll
        for (int k = 0; k < p; ++k) s -= a[p][k] * a[p][k];
        if (!(s > 0.0)) { ok = false; s = 1.0; }
        const double idg = upr_rsqrt(s);
        a[p][p] = idg;   // the diagonal holds the reciprocal pivot
#pragma unroll
        for (int i = p + 1; i < n; ++i) {
            double v = a[i][p];
#pragma unroll
            for (int k = 0; k < p; ++k) v -= a[i][k] * a[p][k];
            a[i][p] = v * idg;
        }
    }
    // inverse, column by column
#pragma unroll
    for (int j = 0; j < n; ++j) {
        double c[n];
        c[j] = a[j][j];
#pragma unroll
        for (int i = j + 1; i < n; ++i) {
            double v = 0.0;
#pragma unroll
            for (int k = j; k < i; ++k) v += a[i][k] * c[k];
            c[i] = -v * a[i][i];
        }
#pragma unroll
        for (int i = 0; i < n; ++i) Li[i * n + j] = (i >= j) ? c[i] : 0.0;
    }
    return ok;
}

// Wave-synchronous Cholesky + triangular inverse in LDS: executed by ONE wave (wc.nt lanes), no
// workgroup barrier inside.  M (n x n, lower triangle used) is overwritten by L; Li receives L^-1.
static inline UPR_HD void upr_chol_inv_wave(const upr_ctx& wc, double* M, double* Li, int n, double* flag) {
    const upr_ctx& ctx = wc;
    for (int p = 0; p < n; ++p) {
        const double piv = M[p * n + p];
        if (ctx.tid == 0 && !(piv > 0.0)) flag[0] = 1.0;
        const double idg = 1.0 / sqrt(piv > 0.0 ? piv : 1.0);
        UPR_WSYNC();
        UPR_FOR(i, n) { if (i > p) M[i * n + p] *= idg; else if (i == p) M[p * n + p] = piv * idg; }
        UPR_WSYNC();
        const int m = n - p - 1;
        UPR_FOR(e, m * m) {
            const int i = p + 1 + e / m, j2 = p + 1 + e % m;
            if (j2 <= i) M[i * n + j2] -= M[i * n + p] * M[j2 * n + p];
        }
        UPR_WSYNC();
    }
    UPR_FOR(j2, n) {
        for (int i = 0; i < j2; ++i) Li[i * n + j2] = 0.0;
        Li[j2 * n + j2] = 1.0 / M[j2 * n + j2];
        for (int i = j2 + 1; i < n; ++i) {
            double v = 0.0;
            for (int k = j2; k < i; ++k) v += M[i * n + k] * Li[k * n + j2];
            Li[i * n + j2] = -v / M[i * n + i];
        }
    }
    UPR_WSYNC();
}

template <class D>
struct upr_qp2 {
    upr_ctx ctx;
    const upr_problem* P;
    int N, neN, n1;
    upr_qp2_ws<D> w;
    upr_qp2_lds<D> o;
    double* L;
    const double* xs; const double* us; const double* x0; const double* lin; const double* Dfg;
    double* ws;
    int lin_stride, lin_g, lin_gx, lin_grad, lin_hess;
    double sigma_mu;
    int mode;  // 0 predictor, 1 corrector (build targets), 2 Lagrangian gradient, 3 corrector (stored targets)

    UPR_HD upr_qp2(const upr_ctx& c, const upr_qp_args& A, int b, double* lds)
        : ctx(c), P(A.P), N(A.d.N), neN(A.d.neN), n1(A.d.N + 1), w(A.d.N, A.d.neN), o(A.d.N, c.nt), L(lds) {
        xs = A.xs + (size_t)b * n1 * D::NX; us = A.us + (size_t)b * N * D::NU; x0 = A.x0 + (size_t)b * D::NX;
        lin = A.lin + (size_t)b * n1 * A.d.lin_stride; Dfg = A.Df + (size_t)b * D::NE * D::NFC;
        ws = A.ws + (size_t)b * A.d.ws_stride;
        lin_stride = A.d.lin_stride; lin_g = A.d.lin_g; lin_gx = A.d.lin_gx; lin_grad = A.d.lin_grad; lin_hess = A.d.lin_hess;
        sigma_mu = 0.0; mode = 0;
    }

    UPR_HD double* Zx(int k) const { return L + o.Z + k * D::NX; }
    UPR_HD double* Zu(int k) const { return L + o.Z + n1 * D::NX + k * D::NU; }
    UPR_HD double* Sx(int k) const { return L + o.S + k * D::NX; }
    UPR_HD double* Su(int k) const { return L + o.S + n1 * D::NX + k * D::NU; }
    UPR_HD double* pre(int k) const { return ws + w.pre + (size_t)k * D::PR_STRIDE; }
    UPR_HD double* store(int k) const { return ws + w.store + (size_t)k * D::SS_STRIDE; }
    UPR_HD const double* rec(int k) const { return lin + (size_t)k * lin_stride; }

    // ---- inequality rows ------------------------------------------------------------------------------
    static UPR_HD bool active(int N_, int k, int j) { return (j < 2 * D::NX) ? (k >= 1) : (k < N_); }
    UPR_HD double row_dot(int j, const double* X, const double* U, bool with_bounds) const {
        // with_bounds: c_j(X, U); else G_j . (X, U)
        if (j < D::NX) return X[j] - (with_bounds ? P->x_lb[j] : 0.0);
        j -= D::NX;
        if (j < D::NX) return (with_bounds ? P->x_ub[j] : 0.0) - X[j];
        j -= D::NX;
        if (j < D::NU) return U[j] - (with_bounds ? P->u_lb[j] : 0.0);
        j -= D::NU;
        if (j < D::NU) return (with_bounds ? P->u_ub[j] : 0.0) - U[j];
        j -= D::NU;
        const int ci = j / 5, r = j % 5;
        double e[3];
        upr_friction_row_jac(P, ci, r, e);
        const double* f = U + D::NQ + 3 * ci;
        return e[0] * f[0] + e[1] * f[1] + e[2] * f[2];
    }

    // ---- flat phase 1: per-inequality weights and targets ---------------------------------------------
    UPR_HD void ineq_prepare() {
        double* t = ws + w.t; double* lam = ws + w.lam; double* rcg = ws + w.rc; double* wq = ws + w.wq; double* sv = ws + w.sv;
        UPR_FOR(e, n1 * D::NI) {
            const int k = e / D::NI, j = e % D::NI;
            double s = 0.0, wgt = 0.0;
            if (active(N, k, j)) {
                const double tj = t[e], lj = lam[e];
                const double c = row_dot(j, Zx(k), Zu(k < N ? k : 0), true);
                const double rp = c - tj;
                wgt = lj / tj;
                if (mode == 0) s = wgt * rp;
                else if (mode == 2) s = -lj;
                else {
                    double rc;
                    if (mode == 1) {
                        const double dta = row_dot(j, Sx(k), Su(k < N ? k : 0), false) + rp;
                        const double dla = -lj - wgt * dta;
                        rc = lj * tj + dta * dla - sigma_mu;
                        rcg[e] = rc;
                    } else rc = rcg[e];
                    s = (rc + lj * rp) / tj - lj;
                }
            }
            wq[e] = wgt; sv[e] = s;
        }
        UPR_SYNC();
    }

    // ---- flat phase 2: reduced gradients, barrier diagonals, residuals of dynamics and equality ---------
    UPR_HD void stage_vectors() {
        const double h = P->dt, h2 = 0.5 * h * h, h3 = h * h * h / 6.0;
        const double* wq = ws + w.wq; const double* sv = ws + w.sv;
        const double* dx = ws + w.dx;
        UPR_FOR(e, n1 * D::NX) {
            const int k = e / D::NX, i = e % D::NX;
            const double* s = sv + (size_t)k * D::NI; const double* q = wq + (size_t)k * D::NI;
            double g = 0.0;
            if (k < N) {
                g = P->Qdiag[i] * (Zx(k)[i] - P->xd[i]);
                if (i < D::NQ) {
                    const double* r = rec(k);
                    double a = r[lin_grad + i];
                    for (int j = 0; j < D::NQ; ++j) a += r[lin_hess + upr_tri(D::NQ, i, j)] * dx[k * D::NX + j];
                    g += a;
                }
                g *= h;
            }
            double* pr = pre(k);
            pr[D::PR_GXS + i] = g + s[i] - s[D::NX + i];
            pr[D::PR_WX + i] = q[i] + q[D::NX + i];
        }
        UPR_FOR(e, N * D::NU) {
            const int k = e / D::NU, i = e % D::NU;
            const double* s = sv + (size_t)k * D::NI + 2 * D::NX; const double* q = wq + (size_t)k * D::NI + 2 * D::NX;
            double g = h * P->Rdiag[i] * Zu(k)[i] + s[i] - s[D::NU + i];
            if (D::NP > 0 && i >= D::NQ) {
                const int fi = i - D::NQ, ci = fi / 3, a = fi % 3;
                for (int r = 0; r < 5; ++r) {
                    double e3[3];
                    upr_friction_row_jac(P, ci, r, e3);
                    g += e3[a] * s[2 * D::NU + 5 * ci + r];
                }
            }
            double* pr = pre(k);
            pr[D::PR_GUS + i] = g;
            if (i < D::NQ) pr[D::PR_WUJ + i] = q[i] + q[D::NU + i];
        }
        UPR_FOR(e, N * D::NQ) {
            const int k = e / D::NQ, j = e % D::NQ;
            const double* X = Zx(k); const double* Xn = Zx(k + 1); const double* U = Zu(k);
            const double q = X[j], v = X[D::NQ + j], a = X[2 * D::NQ + j], u = U[j];
            double* pr = pre(k);
            pr[D::PR_BK + j] = q + h * v + h2 * a + h3 * u - Xn[j];
            pr[D::PR_BK + D::NQ + j] = v + h * a + h2 * u - Xn[D::NQ + j];
            pr[D::PR_BK + 2 * D::NQ + j] = a + h * u - Xn[2 * D::NQ + j];
        }
        const double* du = ws + w.du;
        UPR_FOR(e, N * D::NE) {
            const int k = e / D::NE, r = e % D::NE;
            const double* rc_ = rec(k);
            double v = rc_[lin_g + r];
            for (int j = 0; j < D::NX; ++j) v += rc_[lin_gx + r * D::NX + j] * dx[k * D::NX + j];
            for (int j = 0; j < D::NFC; ++j) v += Dfg[r * D::NFC + j] * du[k * D::NU + D::NQ + j];
            pre(k)[D::PR_EK + r] = v;
        }
        UPR_SYNC();
    }

    // ---- flat phase 3 (factorisation only): contact blocks, Schur complement, Vc ------------------------
    UPR_HD void stage_factors() {
        const double h = P->dt;
        const double* wq = ws + w.wq;
        UPR_FOR(e, N * D::NC) {
            const int k = e / D::NC, ci = e % D::NC;
            const double* q = wq + (size_t)k * D::NI;
            double* pr = pre(k);
            if (D::NF == 3) {
                double Hc[9];
                for (int a = 0; a < 9; ++a) Hc[a] = 0.0;
                for (int a = 0; a < 3; ++a) {
                    const int i = D::NQ + 3 * ci + a;
                    Hc[4 * a] = h * P->Rdiag[i] + q[2 * D::NX + i] + q[2 * D::NX + D::NU + i];
                }
                for (int r = 0; r < 5; ++r) {
                    double e3[3];
                    upr_friction_row_jac(P, ci, r, e3);
                    const double wg = q[2 * D::NX + 2 * D::NU + 5 * ci + r];
                    for (int a = 0; a < 3; ++a) for (int b2 = 0; b2 < 3; ++b2) Hc[3 * a + b2] += wg * e3[a] * e3[b2];
                }
                if (!upr_chol_inv3(Hc)) L[o.misc] = 1.0;
                for (int a = 0; a < 9; ++a) pr[D::PR_LFI + 9 * ci + a] = Hc[a];
            } else {
                const int i = D::NQ + ci;
                pr[D::PR_LFI + ci] = 1.0 / sqrt(h * P->Rdiag[i] + q[2 * D::NX + i] + q[2 * D::NX + D::NU + i]);
            }
        }
        UPR_SYNC();
        // S = (Lfi Df')'(Lfi Df') + rho I, lower triangle, flat over (knot, r, c)
        UPR_FOR(e, N * D::NE * D::NE) {
            const int k = e / (D::NE * D::NE), r = (e % (D::NE * D::NE)) / D::NE, c = e % D::NE;
            if (c > r) continue;
            double* pr = pre(k);
            double acc = (r == c) ? upr_qp_rho_s(D::NE, D::NFC) : 0.0;
            for (int i = 0; i < D::NFC; ++i) {
                double zr, zc;
                if (D::NF == 3) {
                    const int ci = i / 3, a = i % 3;
                    const double* Bk = pr + D::PR_LFI + 9 * ci;
                    zr = 0.0; zc = 0.0;
                    for (int b2 = 0; b2 <= a; ++b2) { zr += Bk[3 * a + b2] * Dfg[r * D::NFC + 3 * ci + b2]; zc += Bk[3 * a + b2] * Dfg[c * D::NFC + 3 * ci + b2]; }
                } else { zr = pr[D::PR_LFI + i] * Dfg[r * D::NFC + i]; zc = pr[D::PR_LFI + i] * Dfg[c * D::NFC + i]; }
                acc += zr * zc;
            }
            pr[D::PR_LSI + r * D::NE + c] = acc;
        }
        UPR_SYNC();
        // one lane per knot: Cholesky inverse in place
        UPR_FOR(k, N) {
            double* pr = pre(k);
            if (!upr_chol_inv_serial<D::NE>(pr + D::PR_LSI, pr + D::PR_LSI)) L[o.misc] = 1.0;
        }
        UPR_SYNC();
        UPR_FOR(e, N * D::NE * D::NX) {
            const int k = e / (D::NE * D::NX), rc_ = e % (D::NE * D::NX), r = rc_ / D::NX, c = rc_ % D::NX;
            const double* pr = pre(k); const double* C = rec(k) + lin_gx;
            double v = 0.0;
            for (int m = 0; m <= r; ++m) v += pr[D::PR_LSI + r * D::NE + m] * C[m * D::NX + c];
            pre(k)[D::PR_VC + rc_] = v;
        }
        UPR_SYNC();
    }

    // ---- flat phase 4 (both passes): force / equality part of the back-substitution ----------------------
    //   yf = Lfi guf ; hf = Lfi' yf ; ee = ek - Df hf ; ys = Lsi ee ; cs = C' Lsi' ys
    UPR_HD void stage_force_vectors() {
        UPR_FOR(k, N) {
            double* pr = pre(k);
            double yf[D::NFC > 0 ? D::NFC : 1], hf[D::NFC > 0 ? D::NFC : 1];
            upr_dims dd; dd.nf = D::NF;
            for (int i = 0; i < D::NFC; ++i) yf[i] = upr_blk_lo(dd, pr + D::PR_LFI, pr + D::PR_GUS + D::NQ, i);
            for (int i = 0; i < D::NFC; ++i) hf[i] = upr_blk_up(dd, pr + D::PR_LFI, yf, i);
            double ee[D::NE], ys[D::NE];
            for (int r = 0; r < D::NE; ++r) {
                double v = pr[D::PR_EK + r] + upr_qp_rho_prox(D::NE, D::NFC) * ws[w.nu + k * D::NE + r];
                for (int i = 0; i < D::NFC; ++i) v -= Dfg[r * D::NFC + i] * hf[i];
                ee[r] = v;
            }
            for (int r = 0; r < D::NE; ++r) {
                double v = 0.0;
                for (int m = 0; m <= r; ++m) v += pr[D::PR_LSI + r * D::NE + m] * ee[m];
                ys[r] = v;
            }
            for (int i = 0; i < D::NFC; ++i) pr[D::PR_YF + i] = yf[i];
            for (int r = 0; r < D::NE; ++r) pr[D::PR_YS + r] = ys[r];
        }
        UPR_SYNC();
        UPR_FOR(e, N * D::NX) {
            const int k = e / D::NX, i = e % D::NX;
            const double* pr = pre(k);
            double v = 0.0;
            for (int r = 0; r < D::NE; ++r) v += pr[D::PR_VC + r * D::NX + i] * pr[D::PR_YS + r];
            pre(k)[D::PR_CS + i] = v;
        }
        UPR_SYNC();
    }

    // ---- terminal knot --------------------------------------------------------------------------------
    UPR_HD void terminal(bool mat) {
        const double irho = 1.0 / UPR_QP_RHO_N;
        const double* r = rec(N); const double* pr = pre(N); const double* yN = ws + w.yN;
        const double* dxN = ws + w.dx + N * D::NX;
        // residual of [p_d - p; v; a] + CN dx_N  -> wv[0..neN)
        if (neN > 0) {
            UPR_FOR(q, neN) {
                double v;
                if (q < 3) { v = r[lin_grad + q]; for (int j = 0; j < D::NQ; ++j) v -= r[lin_hess + q * D::NQ + j] * dxN[j]; }
                else v = Zx(N)[D::NQ + (q - 3)];
                L[o.wv + q] = v;
            }
            UPR_SYNC();
        }
        if (mat) {
            UPR_FOR(e, D::NX * D::NX) {
                const int i = e / D::NX, j = e % D::NX;
                double v = (i == j) ? pr[D::PR_WX + i] : 0.0;
                if (neN > 0) {
                    if (i < D::NQ && j < D::NQ) { for (int q = 0; q < 3; ++q) v += irho * r[lin_hess + q * D::NQ + i] * r[lin_hess + q * D::NQ + j]; }
                    else if (i == j) v += irho;
                }
                L[o.Pm + e] = v;
            }
        }
        UPR_FOR(i, D::NX) {
            double v = pr[D::PR_GXS + i];
            if (neN > 0) {
                if (i < D::NQ) { for (int q = 0; q < 3; ++q) v -= r[lin_hess + q * D::NQ + i] * (yN[q] + irho * L[o.wv + q]); }
                else v += yN[3 + (i - D::NQ)] + irho * L[o.wv + 3 + (i - D::NQ)];
            }
            L[o.pv + i] = v;
        }
        UPR_SYNC();
    }

    // ---- sequential backward sweep ----------------------------------------------------------------------
    UPR_HD void backward(bool mat) {
        constexpr int NQ = D::NQ, NX = D::NX;
        const double h = P->dt, h2 = 0.5 * h * h, h3 = h * h * h / 6.0;
        terminal(mat);
        for (int k = N - 1; k >= 0; --k) {
            const double* pr = pre(k); double* st = store(k);
            // stage vectors into LDS
            UPR_FOR(i, NX) { L[o.bk + i] = pr[D::PR_BK + i]; L[o.gxs + i] = pr[D::PR_GXS + i]; L[o.cs + i] = pr[D::PR_CS + i]; L[o.wx + i] = pr[D::PR_WX + i]; }
            UPR_FOR(j, NQ) { L[o.gus + j] = pr[D::PR_GUS + j]; L[o.wuj + j] = pr[D::PR_WUJ + j]; }
            if (mat) UPR_FOR(e, D::NE * NX) L[o.Vc + e] = pr[D::PR_VC + e];
            else {
                UPR_FOR(e, NQ * NX) L[o.V + e] = st[D::SS_V + e];
                UPR_FOR(e, NQ * NQ) L[o.Lji + e] = st[D::SS_LJI + e];
            }
            UPR_SYNC();
            if (mat) {
                // wv = P+ b + p+ ; Tm = B' P+
                UPR_FOR(i, NX) {
                    double pb = 0.0;
                    for (int j = 0; j < NX; ++j) pb += L[o.Pm + i * NX + j] * L[o.bk + j];
                    L[o.wv + i] = L[o.pv + i] + pb;
                    st[D::SS_PB + i] = pb;
                }
                UPR_FOR(e, NQ * NX) {
                    const int j = e / NX, c = e % NX;
                    L[o.Tm + e] = h3 * L[o.Pm + j * NX + c] + h2 * L[o.Pm + (NQ + j) * NX + c] + h * L[o.Pm + (2 * NQ + j) * NX + c];
                }
                UPR_SYNC();
                // A: Hjj = Tm B + R + barrier ; Pm a-columns
                UPR_FOR(e, NQ * NQ) {
                    const int j = e / NQ, m = e % NQ;
                    double v = h3 * L[o.Tm + j * NX + m] + h2 * L[o.Tm + j * NX + NQ + m] + h * L[o.Tm + j * NX + 2 * NQ + m];
                    if (j == m) v += h * P->Rdiag[j] + L[o.wuj + j];
                    L[o.Hjj + e] = v;
                }
                UPR_FOR(e, NX * NQ) { const int i = e / NQ, j = e % NQ; L[o.Pm + i * NX + 2 * NQ + j] += h2 * L[o.Pm + i * NX + j] + h * L[o.Pm + i * NX + NQ + j]; }
                UPR_SYNC();
                // B: Tm a-columns ; Pm v-columns ; one lane factors Hjj
                UPR_FOR(e, NQ * NQ) { const int i = e / NQ, j = e % NQ; L[o.Tm + i * NX + 2 * NQ + j] += h2 * L[o.Tm + i * NX + j] + h * L[o.Tm + i * NX + NQ + j]; }
                UPR_FOR(e, NX * NQ) { const int i = e / NQ, j = e % NQ; L[o.Pm + i * NX + NQ + j] += h * L[o.Pm + i * NX + j]; }
                if (ctx.tid < 64) { upr_ctx wc; wc.tid = ctx.tid; wc.nt = ctx.nt < 64 ? ctx.nt : 64; upr_chol_inv_wave(wc, L + o.Hjj, L + o.Lji, NQ, L + o.misc); }
                UPR_SYNC();
                // C: Tm v-columns ; Pm a-rows
                UPR_FOR(e, NQ * NQ) { const int i = e / NQ, j = e % NQ; L[o.Tm + i * NX + NQ + j] += h * L[o.Tm + i * NX + j]; }
                UPR_FOR(e, NQ * NX) { const int j = e / NX, c = e % NX; L[o.Pm + (2 * NQ + j) * NX + c] += h2 * L[o.Pm + j * NX + c] + h * L[o.Pm + (NQ + j) * NX + c]; }
                UPR_SYNC();
                // D: Pm v-rows ; V = Lji Hux
                UPR_FOR(e, NQ * NX) { const int j = e / NX, c = e % NX; L[o.Pm + (NQ + j) * NX + c] += h * L[o.Pm + j * NX + c]; }
                UPR_FOR(e, NQ * NX) {
                    const int i = e / NX, c = e % NX;
                    double v = 0.0;
                    for (int m = 0; m <= i; ++m) v += L[o.Lji + i * NQ + m] * L[o.Tm + m * NX + c];
                    L[o.V + e] = v;
                }
                UPR_SYNC();
            } else {
                UPR_FOR(i, NX) L[o.wv + i] = L[o.pv + i] + st[D::SS_PB + i];
                UPR_SYNC();
            }
            // vectors: hx = gxs + A' wv ; huj = gus_j + B' wv
            upr_At_vec(ctx, NQ, h, L + o.wv, L + o.hx);
            UPR_FOR(j, NQ) L[o.huj + j] = L[o.gus + j] + upr_Bt_vec_j(NQ, h, L + o.wv, j);
            UPR_SYNC();
            UPR_FOR(j, NQ) {
                double v = 0.0;
                for (int m = 0; m <= j; ++m) v += L[o.Lji + j * NQ + m] * L[o.huj + m];
                L[o.yj + j] = v;
            }
            UPR_SYNC();
            if (mat) {
                // P = sym(A'P+A) + Q~ - V'V + Vc'Vc   (upper triangle computed, mirrored)
                const double* r = rec(k);
                UPR_FOR(e, NX * NX) {
                    const int i = e / NX, j = e % NX;
                    if (i > j) continue;
                    double v = 0.5 * (L[o.Pm + i * NX + j] + L[o.Pm + j * NX + i]);
                    if (i == j) v += h * P->Qdiag[i] + L[o.wx + i];
                    if (j < NQ) v += h * r[lin_hess + upr_tri(NQ, i, j)];
                    for (int m = 0; m < NQ; ++m) v -= L[o.V + m * NX + i] * L[o.V + m * NX + j];
                    for (int q = 0; q < D::NE; ++q) v += L[o.Vc + q * NX + i] * L[o.Vc + q * NX + j];
                    L[o.Pm + i * NX + j] = v;
                    // mirrored write is deferred to the next phase to avoid reading a half-updated entry
                }
                UPR_FOR(e, NQ * NX) st[D::SS_V + e] = L[o.V + e];
                UPR_FOR(e, NQ * NQ) st[D::SS_LJI + e] = L[o.Lji + e];
            }
            UPR_FOR(i, NX) {
                double v = L[o.hx + i] + L[o.gxs + i] + L[o.cs + i];
                for (int m = 0; m < NQ; ++m) v -= L[o.V + m * NX + i] * L[o.yj + m];
                L[o.pv + i] = v;
            }
            UPR_FOR(j, NQ) st[D::SS_YJ + j] = L[o.yj + j];
            UPR_SYNC();
            if (mat) {
                UPR_FOR(e, NX * NX) { const int i = e / NX, j = e % NX; if (i > j) L[o.Pm + i * NX + j] = L[o.Pm + j * NX + i]; }
                UPR_SYNC();
            }
        }
    }

    // ---- sequential forward sweep (jerk / state recursion), then flat force + multiplier phase -------------
    UPR_HD void forward() {
        constexpr int NQ = D::NQ, NX = D::NX;
        const double h = P->dt, h2 = 0.5 * h * h, h3 = h * h * h / 6.0;
        UPR_FOR(i, NX) Sx(0)[i] = 0.0;
        UPR_SYNC();
        for (int k = 0; k < N; ++k) {
            const double* st = store(k); const double* pr = pre(k);
            UPR_FOR(e, NQ * NX) L[o.V + e] = st[D::SS_V + e];
            UPR_FOR(e, NQ * NQ) L[o.Lji + e] = st[D::SS_LJI + e];
            UPR_FOR(j, NQ) L[o.yj + j] = st[D::SS_YJ + j];
            UPR_FOR(i, NX) L[o.bk + i] = pr[D::PR_BK + i];
            UPR_SYNC();
            const double* sx = Sx(k);
            UPR_FOR(j, NQ) {
                double v = L[o.yj + j];
                for (int c = 0; c < NX; ++c) v += L[o.V + j * NX + c] * sx[c];
                L[o.huj + j] = v;
            }
            UPR_SYNC();
            UPR_FOR(j, NQ) {
                double v = 0.0;
                for (int m = j; m < NQ; ++m) v += L[o.Lji + m * NQ + j] * L[o.huj + m];
                Su(k)[j] = -v;
            }
            UPR_SYNC();
            double* sn = Sx(k + 1); const double* su = Su(k);
            UPR_FOR(j, NQ) {
                const double q = sx[j], v = sx[NQ + j], a = sx[2 * NQ + j], u = su[j];
                sn[j] = q + h * v + h2 * a + h3 * u + L[o.bk + j];
                sn[NQ + j] = v + h * a + h2 * u + L[o.bk + NQ + j];
                sn[2 * NQ + j] = a + h * u + L[o.bk + 2 * NQ + j];
            }
            UPR_SYNC();
        }
        // flat: nu+ = Lsi'(Vc sx + ys) ; su_f = -Lfi'(yf + Lfi Df' nu+) ; terminal multiplier step
        double* nun = ws + w.nun;
        UPR_FOR(k, N) {
            const double* pr = pre(k); const double* sx = Sx(k);
            double t1[D::NE], nu[D::NE];
            for (int r = 0; r < D::NE; ++r) {
                double v = pr[D::PR_YS + r];
                for (int c = 0; c < NX; ++c) v += pr[D::PR_VC + r * NX + c] * sx[c];
                t1[r] = v;
            }
            for (int r = 0; r < D::NE; ++r) {
                double v = 0.0;
                for (int m = r; m < D::NE; ++m) v += pr[D::PR_LSI + m * D::NE + r] * t1[m];
                nu[r] = v; nun[k * D::NE + r] = v;
            }
            double dfn[D::NFC > 0 ? D::NFC : 1], tf[D::NFC > 0 ? D::NFC : 1];
            upr_dims dd; dd.nf = D::NF;
            for (int i = 0; i < D::NFC; ++i) { double v = 0.0; for (int r = 0; r < D::NE; ++r) v += Dfg[r * D::NFC + i] * nu[r]; dfn[i] = v; }
            for (int i = 0; i < D::NFC; ++i) tf[i] = pr[D::PR_YF + i] + upr_blk_lo(dd, pr + D::PR_LFI, dfn, i);
            for (int i = 0; i < D::NFC; ++i) Su(k)[NQ + i] = -upr_blk_up(dd, pr + D::PR_LFI, tf, i);
        }
        if (neN > 0) {
            const double* r = rec(N); const double* dxN = ws + w.dx + N * NX; double* dyN = ws + w.dyN;
            UPR_FOR(q, neN) {
                double v;
                if (q < 3) { v = r[lin_grad + q]; for (int j = 0; j < NQ; ++j) v -= r[lin_hess + q * NQ + j] * (dxN[j] + Sx(N)[j]); }
                else v = Zx(N)[NQ + (q - 3)] + Sx(N)[NQ + (q - 3)];
                dyN[q] = v / UPR_QP_RHO_N;
            }
        }
        UPR_SYNC();
    }

    // ---- costates of the full step: pi+_k = gxs_k + Htilde_xx sx_k + A' pi+_{k+1} + C_k' nu+_k -----------
    UPR_HD void costates() {
        constexpr int NQ = D::NQ, NX = D::NX;
        const double h = P->dt;
        double* pin = ws + w.pin; const double* nun = ws + w.nun; const double* yN = ws + w.yN; const double* dyN = ws + w.dyN;
        // flat part: everything except the A' pi+ recursion
        UPR_FOR(e, n1 * NX) {
            const int k = e / NX, i = e % NX;
            const double* pr = pre(k); const double* sx = Sx(k);
            double v = pr[D::PR_GXS + i] + pr[D::PR_WX + i] * sx[i];
            if (k < N) {
                const double* r = rec(k);
                v += h * P->Qdiag[i] * sx[i];
                if (i < NQ) for (int j = 0; j < NQ; ++j) v += h * r[lin_hess + upr_tri(NQ, i, j)] * sx[j];
                for (int q = 0; q < D::NE; ++q) v += r[lin_gx + q * NX + i] * nun[k * D::NE + q];
            } else if (neN > 0) {
                const double* r = rec(N);
                if (i < NQ) { for (int q = 0; q < 3; ++q) v -= r[lin_hess + q * NQ + i] * (yN[q] + dyN[q]); }
                else v += yN[3 + (i - NQ)] + dyN[3 + (i - NQ)];
            }
            pin[e] = v;
        }
        UPR_SYNC();
        // recursion over the three nq-blocks: independent per joint j
        UPR_FOR(j, NQ) {
            double pq = pin[N * NX + j], pvv = pin[N * NX + NQ + j], pa = pin[N * NX + 2 * NQ + j];
            for (int k = N - 1; k >= 1; --k) {
                const double nq_ = pin[k * NX + j] + pq;
                const double nv_ = pin[k * NX + NQ + j] + h * pq + pvv;
                const double na_ = pin[k * NX + 2 * NQ + j] + 0.5 * h * h * pq + h * pvv + pa;
                pq = nq_; pvv = nv_; pa = na_;
                pin[k * NX + j] = pq; pin[k * NX + NQ + j] = pvv; pin[k * NX + 2 * NQ + j] = pa;
            }
        }
        UPR_SYNC();
    }

    // ---- flat sweeps over all inequality rows with the current step ------------------------------------------
    //   what 0: alpha_max partial ; 1: partial sum (lam + a dlam)(t + a dt) ; 2: apply ; 3: partial max |rp|, aux += lam t
    UPR_HD double ineq_sweep(int what, double alpha, double* aux) {
        double* t = ws + w.t; double* lam = ws + w.lam; const double* rcg = ws + w.rc;
        double acc = (what == 0) ? 1e30 : (what == 5 ? 1e300 : 0.0);   // (what 5: min of the trial products, aux += their sum -- UPR_QP_NGAM)
        UPR_FOR(e, n1 * D::NI) {
            const int k = e / D::NI, j = e % D::NI;
            if (!active(N, k, j)) continue;
            const double tj = t[e], lj = lam[e];
            const double c = row_dot(j, Zx(k), Zu(k < N ? k : 0), true);
            const double rp = c - tj;
            if (what == 3) { const double a = fabs(rp); if (a > acc) acc = a; *aux += lj * tj; continue; }
            const double dt = row_dot(j, Sx(k), Su(k < N ? k : 0), false) + rp;
            const double rc = (mode == 0) ? lj * tj : rcg[e];
            const double dl = -(rc + lj * dt) / tj;
            if (what == 0) {
                if (dt < 0.0) { const double a = -tj / dt; if (a < acc) acc = a; }
                if (dl < 0.0) { const double a = -lj / dl; if (a < acc) acc = a; }
            } else if (what == 1) acc += (lj + alpha * dl) * (tj + alpha * dt);
            else if (what == 5) { const double v = (lj + alpha * dl) * (tj + alpha * dt); acc = fmin(acc, v); *aux += v; }
            else { t[e] = tj + alpha * dt; lam[e] = lj + alpha * dl; }
        }
        UPR_SYNC();
        return acc;
    }

    // ---- explicit KKT residuals (flat) ---------------------------------------------------------------------------
    UPR_HD void residuals(int ntot, double* res) {
        constexpr int NQ = D::NQ, NX = D::NX;
        const double h = P->dt;
        const double* pi = ws + w.pi; const double* nu = ws + w.nu; const double* yN = ws + w.yN;
        const int save = mode;
        mode = 2;
        ineq_prepare();
        stage_vectors();
        mode = save;
        double r_stat = 0.0, r_eq = 0.0;
        UPR_FOR(e, N * NX) {  // knots 1..N
            const int k = 1 + e / NX, i = e % NX;
            const double* pr = pre(k);
            double v = pr[D::PR_GXS + i] - pi[k * NX + i];
            if (k < N) {
                const double* pn = pi + (k + 1) * NX; const double* r = rec(k);
                const int blk = i / NQ, j = i % NQ;
                v += pn[j] * (blk == 0 ? 1.0 : (blk == 1 ? h : 0.5 * h * h));
                if (blk >= 1) v += pn[NQ + j] * (blk == 1 ? 1.0 : h);
                if (blk == 2) v += pn[2 * NQ + j];
                for (int q = 0; q < D::NE; ++q) v += r[lin_gx + q * NX + i] * nu[k * D::NE + q];
            } else if (neN > 0) {
                const double* r = rec(N);
                if (i < NQ) { for (int q = 0; q < 3; ++q) v -= r[lin_hess + q * NQ + i] * yN[q]; }
                else v += yN[3 + (i - NQ)];
            }
            r_stat = fmax(r_stat, fabs(v));
        }
        UPR_FOR(e, N * D::NU) {
            const int k = e / D::NU, i = e % D::NU;
            const double* pr = pre(k);
            double v = pr[D::PR_GUS + i];
            if (i < NQ) v += upr_Bt_vec_j(NQ, h, pi + (k + 1) * NX, i);
            else for (int q = 0; q < D::NE; ++q) v += Dfg[q * D::NFC + (i - NQ)] * nu[k * D::NE + q];
            r_stat = fmax(r_stat, fabs(v));
        }
        UPR_FOR(e, N * NX) { const int k = e / NX, i = e % NX; r_eq = fmax(r_eq, fabs(pre(k)[D::PR_BK + i])); }
        UPR_FOR(e, N * D::NE) { const int k = e / D::NE, r = e % D::NE; r_eq = fmax(r_eq, fabs(pre(k)[D::PR_EK + r])); }
        if (neN > 0) {
            const double* r = rec(N); const double* dxN = ws + w.dx + N * NX;
            UPR_FOR(q, neN) {
                double v;
                if (q < 3) { v = r[lin_grad + q]; for (int j = 0; j < NQ; ++j) v -= r[lin_hess + q * NQ + j] * dxN[j]; }
                else v = Zx(N)[NQ + (q - 3)];
                r_eq = fmax(r_eq, fabs(v));
            }
        }
        double lt = 0.0;
        double r_in = ineq_sweep(3, 0.0, &lt);
        res[0] = upr_reduce(ctx, L + o.red, r_stat, 1);
        res[1] = upr_reduce(ctx, L + o.red, r_eq, 1);
        res[2] = upr_reduce(ctx, L + o.red, r_in, 1);
        res[3] = upr_reduce(ctx, L + o.red, lt, 0) / (ntot > 0 ? ntot : 1);
    }

    // ---- driver ------------------------------------------------------------------------------------------------------
    double* prof;
    long long tlast;
    UPR_HD void tic() {
#ifndef UPR_HOST_EMU
        if (prof && ctx.tid == 0) tlast = (long long)__builtin_readcyclecounter();
#endif
    }
    UPR_HD void toc(int id) {
#ifndef UPR_HOST_EMU
        if (prof && ctx.tid == 0) { long long t = (long long)__builtin_readcyclecounter(); prof[id] += (double)(t - tlast); tlast = t; }
#endif
    }
    UPR_HD void solve(double* stats_b) {
        constexpr int NX = D::NX, NU = D::NU;
        tic();
        UPR_FOR(i, w.pre) ws[i] = 0.0;
        if (ctx.tid == 0) L[o.misc] = 0.0;
        UPR_SYNC();
        UPR_FOR(i, NX) ws[w.dx + i] = x0[i] - xs[i];
        UPR_FOR(e, n1 * NX) { const int k = e / NX; L[o.Z + e] = (k == 0) ? x0[e] : xs[e]; L[o.S + e] = 0.0; }
        UPR_FOR(e, N * NU) { L[o.Z + n1 * NX + e] = us[e]; L[o.S + n1 * NX + e] = 0.0; }
        UPR_SYNC();
        UPR_FOR(e, n1 * D::NI) {
            const int k = e / D::NI, j = e % D::NI;
            double t = 1.0, lam = 0.0;
            if (active(N, k, j)) {
                const double c = row_dot(j, Zx(k), Zu(k < N ? k : 0), true);
                t = c > UPR_QP_THR ? c : UPR_QP_THR;
                lam = UPR_QP_MU0 / t;
            }
            ws[w.t + e] = t; ws[w.lam + e] = lam;
        }
        UPR_SYNC();
        const int ntot = N * (2 * NU + D::NP) + N * 2 * NX;
        double res[4] = {0, 0, 0, 0};
        int it = 0, status = 1;
        const double tol = P->qp_tol;
        const double tol_stat = P->qp_tol_stat > 0.0 ? P->qp_tol_stat : tol;   // HPIPM tol_stat (upright_mi.h)
        for (;; ++it) {
            residuals(ntot, res);
            toc(0);
#ifdef UPR_HOST_EMU
            if (getenv("UPR_EMU_DEBUG")) printf("v2 it %d res %.3e %.3e %.3e %.3e\n", it, res[0], res[1], res[2], res[3]);
#endif
            if (it > 0 && res[0] < tol_stat && res[1] < tol && res[2] < tol && res[3] < tol) { status = 0; break; }
            if (it >= P->qp_iter_max) break;
            const double mu = res[3];
            // predictor
            mode = 0;
            ineq_prepare(); toc(1); stage_vectors(); toc(2); stage_factors(); toc(3); stage_force_vectors(); toc(4);
            backward(true); toc(5);
            if (L[o.misc] != 0.0) { status = 2; break; }
            forward(); toc(6);
            double a_aff = upr_reduce(ctx, L + o.red, ineq_sweep(0, 0.0, nullptr), 2);
            if (a_aff > 1.0) a_aff = 1.0;
            const double mu_aff = upr_reduce(ctx, L + o.red, ineq_sweep(1, a_aff, nullptr), 0) / ntot;
            const double sg = mu_aff / mu;
            sigma_mu = sg * sg * sg * mu;
            if (sigma_mu < UPR_QP_SIGMA_FLOOR * tol) sigma_mu = UPR_QP_SIGMA_FLOOR * tol;
            toc(7);
            // corrector
            mode = 1;
            ineq_prepare(); toc(1); stage_vectors(); toc(2); stage_force_vectors(); toc(4);
            backward(false); toc(8);
            forward(); toc(6);
            mode = 3;
            costates(); toc(9);
            double a = upr_reduce(ctx, L + o.red, ineq_sweep(0, 0.0, nullptr), 2);
            if (a > 1.0) a = 1.0;
            a *= 0.995;   // see upr_qp.h
            if (UPR_QP_NGAM > 0.0 && it < UPR_QP_NIT) {   // centrality safeguard (see UPR_QP_NGAM)
                double sm = 0.0;
                const double mn = upr_reduce(ctx, L + o.red, ineq_sweep(5, a, &sm), 2);
                sm = upr_reduce(ctx, L + o.red, sm, 0);
                if (!(mn >= UPR_QP_NGAM * (sm / (ntot > 0 ? ntot : 1)))) a *= UPR_QP_NBT;
            }
            ineq_sweep(2, a, nullptr);
            UPR_FOR(e, n1 * NX) {
                const double s = L[o.S + e];
                if (e >= NX) { L[o.Z + e] += a * s; ws[w.dx + e] += a * s; }
                ws[w.pi + e] += a * (ws[w.pin + e] - ws[w.pi + e]);
            }
            UPR_FOR(e, N * NU) { const double s = L[o.S + n1 * NX + e]; L[o.Z + n1 * NX + e] += a * s; ws[w.du + e] += a * s; }
            UPR_FOR(e, N * D::NE) ws[w.nu + e] += a * (ws[w.nun + e] - ws[w.nu + e]);
            UPR_FOR(e, neN) ws[w.yN + e] += a * ws[w.dyN + e];
            UPR_SYNC();
            toc(10);
        }
        if (ctx.tid == 0) { stats_b[1] = it; stats_b[2] = status; stats_b[6] = res[0]; stats_b[7] = res[1]; stats_b[8] = res[2]; stats_b[9] = res[3]; }
        UPR_SYNC();
    }
};

template <class D>
static inline UPR_HD void upr_qp2_solve(const upr_ctx& ctx, const upr_qp_args& A, int b, double* L) {
    upr_qp2<D> S(ctx, A, b, L);
    S.prof = A.prof ? A.prof + (size_t)b * 16 : nullptr;
    S.solve(A.stats + (size_t)b * UPR_NSTATS);
    if (ctx.tid == 0) upr_qp_store_key(A, b, (int)A.stats[(size_t)b * UPR_NSTATS + 1]);
}

template <class D>
static inline UPR_HD size_t upr_qp2_ws_doubles(int N, int neN) { return (size_t)upr_qp2_ws<D>(N, neN).total; }
template <class D>
static inline UPR_HD size_t upr_qp2_lds_doubles(int N, int nt) { return (size_t)upr_qp2_lds<D>(N, nt).total; }

#ifndef UPR_HOST_EMU
template <class D, int NT>
__global__ void __launch_bounds__(NT, 4) upr_qp2_kernel(upr_qp_args A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    upr_ctx ctx; ctx.tid = threadIdx.x; ctx.nt = NT;
    upr_qp2_solve<D>(ctx, A, upr_qp_instance(A, blockIdx.x), smem);
}
#endif
