// upr_linesearch.h -- merit evaluation, filter line search and SQP convergence test, one workgroup per
// instance, one lane per shooting knot.
//
// What it replaces: [UPSTREAM, absent] ocs2_sqp MultipleShootingSolver::takeStep / computePerformance /
// checkConvergence with the ocs2_sqp defaults (alpha_decay 0.5, alpha_min 1e-4, gamma_c 1e-6,
// g_max 1e6, g_min 1e-6, armijo 1e-4; none is bound at upright_control/src/pybindings.cpp:190-213).
// The per-knot terms are the OCP terms of controller_interface.cpp:136-357 re-evaluated (values only)
// at the trial iterate: quadratic state-input cost, end-effector cost, object-dynamics equality,
// friction-cone rows, boxes, dynamics defects, terminal equality.
#pragma once
#include "upr_kin.h"

struct upr_ls_args {
    const upr_problem* P;
    upr_dims d;
    double* xs;            // [B][N+1][nx]  (updated in place when a step is accepted)
    double* us;            // [B][N][nu]
    const double* x0;      // [B][nx]
    const double* t0;      // [B]
    const double* body_params;
    const double* way_p;
    const double* lin;     // for the Armijo descent metric (cost gradient at the linearisation point)
    const double* ws;      // dx, du live in the QP workspace
    double* stats;         // [B][UPR_NSTATS]
    int* done;             // [B] convergence flag
    int iter;              // SQP iteration index (0-based)
    const double* dyn = nullptr;    // [B][n_dyn][9] observed dynamic-obstacle states (NULL: none)
    const double* pflag = nullptr;  // [B] projectile activation flag
    const double* way_q = nullptr;  // [B][n_way][4] target orientations; NULL unless Wee[3..5] != 0
    // Engine bookkeeping folded into this launch (device kernel only; each was a stream operation of its own):
    //   order_out: the dispatch order of the NEXT QP launch, longest first by the iteration count the QP that just ran left in
    //              iter_key[.] (one byte per instance, upr_qp_args::iter_key) -- every workgroup ranks its own instance (ties by index);
    //   xs_prev ..: the solution remembered for the next warm start / policy evaluation (the advance's last line search)
    int* order_out = nullptr;
    const unsigned char* iter_key = nullptr;
    double* xs_prev = nullptr; double* us_prev = nullptr; double* tprev = nullptr;
    int stage_full = 1;    // the instance's trajectory and step staged in LDS (upr_ls_lds_doubles)
};

// performance terms of knot k at step length alpha: out += [cost, dyn_sse, eq_sse, ineq_sse]
// NFM / NBM: compile-time bounds of nf nc / nb (the per-lane input vector and body wrenches stay in registers for the
// small shapes: with the library-wide maxima they lived in scratch, 2 KB per lane)
// EXACT: the problem has exactly nf = 3, nc = NFM / 3 contacts, nb = NBM bodies (checked by the launcher; OBS: with collision / projectile rows -- round 5, the thrown-ball shape): every
// loop bound below is then a compile-time constant, the trial state, input and wrenches stay in registers (with
// run-time bounds they were indexed dynamically: 1.2 KB of scratch per lane)
// contact wrench on the single balanced body (contact_constraints.h:107-157 with nb = 1: every contact joins tray and body)
template <int NCX>
static UPR_HDI void upr_object_wrench_single(const upr_problem* P, const double* bp, const double* forces, double* W) {
    const double com[3] = {bp[1] / bp[0], bp[2] / bp[0], bp[3] / bp[0]};
#pragma unroll
    for (int i = 0; i < 6; ++i) W[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NCX; ++i) {
        const double f0 = forces[3 * i], f1 = forces[3 * i + 1], f2 = forces[3 * i + 2];
        const double l0 = P->contact_r2[i][0] - com[0], l1 = P->contact_r2[i][1] - com[1], l2 = P->contact_r2[i][2] - com[2];
        W[0] -= f0; W[1] -= f1; W[2] -= f2;
        W[3] -= l1 * f2 - l2 * f1; W[4] -= l2 * f0 - l0 * f2; W[5] -= l0 * f1 - l1 * f0;
    }
}
template <int NQ, int NFM = 3 * UPR_MAX_CONTACTS, int NBM = UPR_MAX_BODIES, bool EXACT = false, bool OBS = !EXACT>
// Xt / Ut: the trial trajectory xs + alpha dx, us + alpha du of the instance (staged by the workgroup, LDS on the device);
// sc: (sin, cos) of the trial joint angles of every knot, [N + 1][NQ][2]
static UPR_HDI void upr_ls_knot(const upr_ls_args& A, int b, int k, const double* Xt, const double* Ut, const double* sc, double* out) {
    const upr_problem* P = A.P; const upr_dims& d = A.d;
    const int N = d.N;
    constexpr int nq = NQ, nx = 3 * NQ;
    const int nu = EXACT ? NQ + NFM : d.nu;
    const double h = P->dt, h2 = 0.5 * h * h, h3 = h * h * h / 6.0;
    double X[3 * NQ], U[NQ + NFM];
#pragma unroll
    for (int i = 0; i < nx; ++i) X[i] = Xt[k * nx + i];
    double cost = 0.0, dyn = 0.0, eq = 0.0, iq = 0.0;
    const double wt = (k < N) ? h : 1.0;
    if (k == 0) {
#pragma unroll
        for (int i = 0; i < nx; ++i) { double e = A.x0[(size_t)b * nx + i] - X[i]; dyn += e * e; }
    }
    if (k >= 1) {
#pragma unroll
        for (int i = 0; i < nx; ++i) {
            double v = fmin(0.0, fmin(X[i] - P->x_lb[i], P->x_ub[i] - X[i]));
            iq += wt * v * v;
        }
    }
    if (OBS && d.no > 0 && k >= 1 && k < N) {   // collision rows (knots 1..N-1); !OBS: the problem has none
        double dd[UPR_MAX_PAIRS + 8], xo[9 * UPR_MAX_DYN];
        if (A.dyn) for (int oi = 0; oi < P->n_dyn; ++oi) upr_obstacle_at(A.dyn + ((size_t)b * P->n_dyn + oi) * 9, k * h, xo + 9 * oi, xo + 9 * oi + 3, xo + 9 * oi + 6);
        upr_obstacle_values<NQ>(P, X, A.dyn ? xo : nullptr, A.pflag ? A.pflag[b] : 0.0, dd);
        for (int r = 0; r < d.no; ++r) { const double v = fmin(0.0, dd[r]); iq += h * v * v; }
    }
    upr_ee<double> E;
    upr_ee_kinematics<double, NQ>(P, X, -1, E, sc + k * 2 * NQ);
    double pd[3];
    upr_target_position(P, A.way_p + (size_t)b * P->n_way * 3, A.t0[b] + k * h, pd);
    if (k < N) {
#pragma unroll
        for (int i = 0; i < (EXACT ? NQ + NFM : nu); ++i) U[i] = Ut[k * nu + i];
        double c = 0.0;
#pragma unroll
        for (int i = 0; i < nx; ++i) { double e = X[i] - P->xd[i]; c += 0.5 * P->Qdiag[i] * e * e; }
#pragma unroll
        for (int i = 0; i < (EXACT ? NQ + NFM : nu); ++i) c += 0.5 * P->Rdiag[i] * U[i] * U[i];
        for (int r = 0; r < 3; ++r) { double e = E.p[r] - pd[r]; c += 0.5 * P->Wee[r] * e * e; }
        if (A.way_q) {
            double Rr[9], eo[3];
            upr_target_rotation(P, A.way_q + (size_t)b * P->n_way * 4, A.t0[b] + k * h, Rr);
            upr_orientation_error<double>(E.C, Rr, eo);
            for (int r = 0; r < 3; ++r) c += 0.5 * P->Wee[3 + r] * eo[r] * eo[r];
        }
        cost += h * c;
        // dynamics defect against the next trial state
#pragma unroll
        for (int j = 0; j < nq; ++j) {
            double q = X[j], v = X[nq + j], a = X[2 * nq + j], u = U[j];
            const double* xn = Xt + (k + 1) * nx;
            double e0 = q + h * v + h2 * a + h3 * u - xn[j];
            double e1 = v + h * a + h2 * u - xn[nq + j];
            double e2 = a + h * u - xn[2 * nq + j];
            dyn += h * (e0 * e0 + e1 * e1 + e2 * e2);
        }
        // object-dynamics equality
        double Fw[6 * NBM];
        const int nb = EXACT ? NBM : d.nb;
        const double* bp = A.body_params + (size_t)b * nb * 10;
        if (EXACT && NBM == 1) upr_object_wrench_single<NFM / 3>(P, bp, U + nq, Fw);
        else upr_object_wrenches(P, bp, U + nq, Fw);
        const double sc = A.d.eq_scale;
#pragma unroll
        for (int bb = 0; bb < (EXACT ? NBM : nb); ++bb) {
            double g[6];
            upr_body_residual<double>(E, bp + 10 * bb, P->gravity, Fw + 6 * bb, Fw + 6 * bb + 3, g);
            for (int r = 0; r < 6; ++r) eq += h * (sc * g[r]) * (sc * g[r]);
        }
        // friction rows and input box
        if (EXACT || d.np > 0) {
#pragma unroll
            for (int ci = 0; ci < (EXACT ? NFM / 3 : d.nc); ++ci) {
                double hr[5];
                upr_friction_rows_contact(P, ci, U + nq + 3 * ci, hr);
                for (int r = 0; r < 5; ++r) { double v = fmin(0.0, hr[r]); iq += h * v * v; }
            }
        }
#pragma unroll
        for (int i = 0; i < (EXACT ? NQ + NFM : nu); ++i) {
            double v = fmin(0.0, fmin(U[i] - P->u_lb[i], P->u_ub[i] - U[i]));
            iq += h * v * v;
        }
    } else if (d.neN > 0) {
        for (int r = 0; r < 3; ++r) { double e = pd[r] - E.p[r]; eq += e * e; }
#pragma unroll
        for (int i = 0; i < 2 * nq; ++i) eq += X[nq + i] * X[nq + i];
    }
    out[0] += cost; out[1] += dyn; out[2] += eq; out[3] += iq;
}

// the same terms at the CURRENT iterate (alpha = 0), where the linearisation kernel has just been: the end-effector cost, the
// object-dynamics residual, the collision rows and the terminal position error are read out of the knot's record instead
// of walking the chain again
template <int NQ, int NFM = 3 * UPR_MAX_CONTACTS, bool EXACT = false, bool OBS = !EXACT>
static UPR_HDI void upr_ls_knot_base(const upr_ls_args& A, int b, int k, const double* xs_l, const double* us_l, double* out) {
    const upr_problem* P = A.P; const upr_dims& d = A.d;
    const int N = d.N;
    constexpr int nq = NQ, nx = 3 * NQ;
    const int nu = EXACT ? NQ + NFM : d.nu;
    const double h = P->dt, h2 = 0.5 * h * h, h3 = h * h * h / 6.0;
    const double* X = xs_l + k * nx;   // (the instance's trajectory, staged)
    const double* rec = A.lin + ((size_t)b * (N + 1) + k) * d.lin_stride;
    double cost = 0.0, dyn = 0.0, eq = 0.0, iq = 0.0;
    const double wt = (k < N) ? h : 1.0;
    // (compile-time trip counts: the loads of a loop are requested together instead of one exposed latency per trip)
    double Xr[3 * NQ];
#pragma unroll
    for (int i = 0; i < nx; ++i) Xr[i] = X[i];
    if (k == 0) {
#pragma unroll
        for (int i = 0; i < nx; ++i) { double e = A.x0[(size_t)b * nx + i] - Xr[i]; dyn += e * e; }
    }
    if (k >= 1) {
#pragma unroll
        for (int i = 0; i < nx; ++i) {
            double v = fmin(0.0, fmin(Xr[i] - P->x_lb[i], P->x_ub[i] - Xr[i]));
            iq += wt * v * v;
        }
    }
    if (OBS && d.no > 0 && k >= 1 && k < N) for (int r = 0; r < d.no; ++r) { const double v = fmin(0.0, rec[d.lin_obs + r]); iq += h * v * v; }
    if (k < N) {
        const double* U = us_l + k * nu;
        const double* xn = X + nx;
        double c = rec[d.lin_cost];
#pragma unroll
        for (int i = 0; i < nx; ++i) { double e = Xr[i] - P->xd[i]; c += 0.5 * P->Qdiag[i] * e * e; }
        double Ur[NQ + NFM];
#pragma unroll
        for (int i = 0; i < (EXACT ? NQ + NFM : nu); ++i) Ur[i] = U[i];
#pragma unroll
        for (int i = 0; i < (EXACT ? NQ + NFM : nu); ++i) c += 0.5 * P->Rdiag[i] * Ur[i] * Ur[i];
        cost += h * c;
#pragma unroll
        for (int j = 0; j < nq; ++j) {
            double q = Xr[j], v = Xr[nq + j], a = Xr[2 * nq + j], u = Ur[j];
            double e0 = q + h * v + h2 * a + h3 * u - xn[j], e1 = v + h * a + h2 * u - xn[nq + j], e2 = a + h * u - xn[2 * nq + j];
            dyn += h * (e0 * e0 + e1 * e1 + e2 * e2);
        }
        for (int r = 0; r < d.ne; ++r) eq += h * rec[d.lin_g + r] * rec[d.lin_g + r];
        if (EXACT || d.np > 0) {
#pragma unroll
            for (int ci = 0; ci < (EXACT ? NFM / 3 : d.nc); ++ci) {
                double hr[5];
                upr_friction_rows_contact(P, ci, Ur + nq + 3 * ci, hr);
                for (int r = 0; r < 5; ++r) { double v = fmin(0.0, hr[r]); iq += h * v * v; }
            }
        }
#pragma unroll
        for (int i = 0; i < (EXACT ? NQ + NFM : nu); ++i) {
            double v = fmin(0.0, fmin(Ur[i] - P->u_lb[i], P->u_ub[i] - Ur[i]));
            iq += h * v * v;
        }
    } else if (d.neN > 0) {
        for (int r = 0; r < 3; ++r) eq += rec[d.lin_grad + r] * rec[d.lin_grad + r];
#pragma unroll
        for (int i = 0; i < 2 * nq; ++i) eq += Xr[nq + i] * Xr[nq + i];
    }
    out[0] += cost; out[1] += dyn; out[2] += eq; out[3] += iq;
}

// block reduction of 4 partials; result broadcast in res[4]
static UPR_HDI void upr_ls_reduce4(const upr_ctx& ctx, double* red, const double* part, double* res) {
    for (int c = 0; c < 4; ++c) red[c * ctx.nt + ctx.tid] = part[c];
    UPR_SYNC();
    for (int s = 1; s < ctx.nt; s <<= 1) {
        if ((ctx.tid & (2 * s - 1)) == 0 && ctx.tid + s < ctx.nt)
            for (int c = 0; c < 4; ++c) red[c * ctx.nt + ctx.tid] += red[c * ctx.nt + ctx.tid + s];
        UPR_SYNC();
    }
    for (int c = 0; c < 4; ++c) res[c] = red[c * ctx.nt];
    UPR_SYNC();
}

#ifndef UPR_HOST_EMU
// one-wave workgroups: the same pairwise tree by cross-lane butterflies (no LDS, no barriers); every lane ends with the sum
static __device__ __forceinline__ void upr_ls_reduce4_wave(const double* part, double* res) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double v = part[c];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
        res[c] = v;
    }
}
#define UPR_LS_REDUCE4(part, res) do { if (ctx.nt == 64) upr_ls_reduce4_wave(part, res); else upr_ls_reduce4(ctx, L, part, res); } while (0)
#else
#define UPR_LS_REDUCE4(part, res) upr_ls_reduce4(ctx, L, part, res)
#endif

// Workgroup scratch (doubles): the reduction area, then the instance's trajectory and step staged with coalesced requests
// (a lane per knot reading its knot's 27 + 21 doubles straight from global memory touches a cache line of its own per
// request: the address unit then serialises 21 lines per instruction -- the baseline pass alone took 24 k cycles), the
// trial trajectory xs + alpha dx, us + alpha du of the step length in work, and (sin, cos) of its joint angles (computed
// a lane per (knot, joint): inside the chain walk they were nine f64 sincos in series per lane)
// (stage_full = 0 -- long horizons of the large shapes, where three copies do not fit: only the trial trajectory and the
// sines / cosines live in LDS, the trajectory and the step are read where they lie)
static UPR_HDI int upr_ls_lds_doubles(const upr_dims& d, int nt, bool full = true) { return 4 * nt + 8 + (full ? 3 : 1) * ((d.N + 1) * d.nx + d.N * d.nu) + 2 * (d.N + 1) * d.nq + 8; }
template <int NQ, int NFM = 3 * UPR_MAX_CONTACTS, int NBM = UPR_MAX_BODIES, bool EXACT = false, bool OBS = !EXACT>
static UPR_HDI void upr_ls_instance(const upr_ctx& ctx, const upr_ls_args& A, int b, double* L) {
    const upr_problem* P = A.P; const upr_dims& d = A.d;
    const int N = d.N, nx = d.nx, nu = d.nu, nq = d.nq;
    if (A.done[b]) return;
    double* st = A.stats + (size_t)b * UPR_NSTATS;
#if defined(UPR_LS_PROF) && !defined(UPR_HOST_EMU)
    // instrumented build (tools/build_prof.sh): cycle stamps of the phases, returned in the QP-residual slots of the statistics
    long long lsp[5]; int lspn = 0;
#define UPR_LS_STAMP() do { __builtin_amdgcn_sched_barrier(0); lsp[lspn++] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define UPR_LS_STAMP() ((void)0)
#endif
    UPR_LS_STAMP();
    const double qp_status = st[2];
    const double alpha_decay = 0.5, alpha_min = 1e-4, gamma_c = 1e-6, g_max = 1e6, g_min = 1e-6, armijo = 1e-4;
    double* xs = A.xs + (size_t)b * (N + 1) * nx; double* us = A.us + (size_t)b * N * nu;
    const double* ws = A.ws + (size_t)b * d.ws_stride;
    const double* dx = ws + d.ws_dx; const double* du = ws + d.ws_du;
    const double* lin = A.lin + (size_t)b * (N + 1) * d.lin_stride;
    const int nxs = (N + 1) * nx, nus = N * nu;
    double* Xt = L + ((4 * ctx.nt + 8 + 1) & ~1); double* Ut = Xt + nxs; double* sc = Ut + nus;
    const double* xs_l = xs; const double* us_l = us; const double* dx_l = dx; const double* du_l = du;
    if (A.stage_full) {
        double* sx_ = sc + 2 * (N + 1) * nq; double* su_ = sx_ + nxs; double* sdx_ = su_ + nus; double* sdu_ = sdx_ + nxs;
        UPR_FOR(i, nxs) { sx_[i] = xs[i]; sdx_[i] = dx[i]; }
        UPR_FOR(i, nus) { su_[i] = us[i]; sdu_[i] = du[i]; }
        UPR_SYNC();
        xs_l = sx_; us_l = su_; dx_l = sdx_; du_l = sdu_;
    }
    // baseline, step norms and Armijo descent metric (cost gradient . step)
    double part[4] = {0, 0, 0, 0}, base[4], aux[4] = {0, 0, 0, 0}, auxr[4];
    // (one pass over the knots for both sets of sums: the requests of the second set are in flight beside the first's)
    UPR_FOR(k, N + 1) {
        upr_ls_knot_base<NQ, NFM, EXACT, OBS>(A, b, k, xs_l, us_l, part);
        constexpr int nxc = 3 * NQ;
        double sx[nxc], xv[nxc], gq[NQ];
#pragma unroll
        for (int i = 0; i < nxc; ++i) { sx[i] = dx_l[k * nxc + i]; xv[i] = (k < N) ? xs_l[k * nxc + i] : 0.0; }
#pragma unroll
        for (int i = 0; i < NQ; ++i) gq[i] = (k < N) ? lin[(size_t)k * d.lin_stride + d.lin_grad + i] : 0.0;
#pragma unroll
        for (int i = 0; i < nxc; ++i) {
            const double s = sx[i];
            aux[1] += s * s;
            if (k < N) {
                double g = P->Qdiag[i] * (xv[i] - P->xd[i]);
                if (i < NQ) g += gq[i];
                aux[0] += P->dt * g * s;
            }
        }
        if (k < N) {
            if (EXACT) {
                constexpr int nuc = NQ + NFM;
                double su[nuc], uv[nuc];
#pragma unroll
                for (int i = 0; i < nuc; ++i) { su[i] = du_l[k * nuc + i]; uv[i] = us_l[k * nuc + i]; }
#pragma unroll
                for (int i = 0; i < nuc; ++i) { aux[2] += su[i] * su[i]; aux[0] += P->dt * P->Rdiag[i] * uv[i] * su[i]; }
            } else for (int i = 0; i < nu; ++i) {
                double s = du_l[k * nu + i];
                aux[2] += s * s;
                aux[0] += P->dt * P->Rdiag[i] * us_l[k * nu + i] * s;
            }
        }
    }
    UPR_LS_REDUCE4(part, base);
    UPR_LS_STAMP();
    UPR_LS_REDUCE4(aux, auxr);
    UPR_LS_STAMP();
    const double descent = auxr[0], dxn = sqrt(auxr[1]), dun = sqrt(auxr[2]);
    const double base_viol = sqrt(base[1] + base[2] + base[3]);
    double alpha = 1.0, perf[4] = {base[0], base[1], base[2], base[3]};
    bool accepted = false;
    if (qp_status != 2.0) {
        do {
            double p2[4] = {0, 0, 0, 0};
            UPR_FOR(i, nxs) Xt[i] = xs_l[i] + alpha * dx_l[i];
            UPR_FOR(i, nus) Ut[i] = us_l[i] + alpha * du_l[i];
            UPR_FOR(e, (N + 1) * NQ) {
                const int k = e / NQ, j = e % NQ;
                double s_ = 0.0, c_ = 1.0;
                if (P->joint_type[j] == 1) upr_sincos(xs_l[k * nx + j] + alpha * dx_l[k * nx + j], &s_, &c_);
                sc[2 * e] = s_; sc[2 * e + 1] = c_;
            }
            UPR_SYNC();
            UPR_FOR(k, N + 1) upr_ls_knot<NQ, NFM, NBM, EXACT, OBS>(A, b, k, Xt, Ut, sc, p2);
            UPR_LS_REDUCE4(p2, perf);
            double viol = sqrt(perf[1] + perf[2] + perf[3]);
            if (viol > g_max) accepted = false;
            else if (viol < g_min) accepted = (descent < 0.0) ? (perf[0] < base[0] + armijo * alpha * descent) : true;
            else accepted = (perf[0] < base[0] - gamma_c * base_viol) || (viol < (1.0 - gamma_c) * base_viol);
            if (accepted) break;
            alpha *= alpha_decay;
        } while (alpha >= alpha_min);
    }
    UPR_LS_STAMP();
    double cost = base[0], viol = base_viol;
    if (accepted) {
        UPR_FOR(i, nxs) xs[i] = Xt[i];   // (the trial trajectory of the accepted step length: xs + alpha dx, as staged)
        UPR_FOR(i, nus) us[i] = Ut[i];
        cost = perf[0]; viol = sqrt(perf[1] + perf[2] + perf[3]);
    }
    bool conv = !accepted;                                                           // STEPSIZE
    if (accepted && fabs(base[0] - cost) < P->cost_tol && viol < g_min) conv = true;  // METRICS
    if (accepted && alpha * dxn < P->delta_tol && alpha * dun < P->delta_tol) conv = true;  // PRIMAL
    if (ctx.tid == 0) {
        st[0] = A.iter + 1; st[3] = accepted ? alpha : 0.0; st[4] = cost; st[5] = viol; st[10] = dxn; st[11] = dun;
        if (conv) A.done[b] = 1;
    }
#if defined(UPR_LS_PROF) && !defined(UPR_HOST_EMU)
    UPR_LS_STAMP();
    if (ctx.tid == 0) for (int i = 0; i < 4; ++i) st[6 + i] = (double)(lsp[i + 1] - lsp[i]);
#endif
    UPR_SYNC();
}

#ifndef UPR_HOST_EMU
template <int NQ, int NT, int NFM = 3 * UPR_MAX_CONTACTS, int NBM = UPR_MAX_BODIES, bool EXACT = false, bool OBS = !EXACT>
__global__ void __launch_bounds__(NT) upr_linesearch_kernel(upr_ls_args A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    upr_ctx ctx; ctx.tid = threadIdx.x; ctx.nt = NT;
    static_assert(NT == 64, "one wave per instance (the rank below is a wave reduction)");
    if (A.order_out) {
        // B one-byte keys, read four at a time by consecutive lanes (B bytes per workgroup out of the L2: 16 MB per launch at
        // B = 4096 where ranking on stats[.][1], one 96-byte stride per instance, moved 1 GB)
        const int B = gridDim.x, me = blockIdx.x;
        const int mk = A.iter_key[me];
        int cnt = 0;
        const unsigned int* k4 = reinterpret_cast<const unsigned int*>(A.iter_key);
        for (int o4 = threadIdx.x; o4 < (B >> 2); o4 += NT) {
            const unsigned int w = k4[o4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int k = (int)((w >> (8 * j)) & 255u), o = 4 * o4 + j; cnt += (k > mk || (k == mk && o < me)) ? 1 : 0; }
        }
        for (int o = (B & ~3) + threadIdx.x; o < B; o += NT) { const int k = A.iter_key[o]; cnt += (k > mk || (k == mk && o < me)) ? 1 : 0; }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
        if (threadIdx.x == 0) A.order_out[cnt] = me;
    }
    // The problem record (joint frames, bounds, weights, contacts: 11 KB) is read all over the evaluation, element by element
    // and mostly on the serial chain walk -- out of global memory every one of those reads is a vector load with a wait of
    // its own (the record may alias the kernel's stores, so they are not scalar loads): a copy in LDS, made with coalesced
    // requests, serves them instead.
    static_assert(sizeof(upr_problem) % sizeof(double) == 0, "copied as doubles");
    constexpr int NPD = (int)(sizeof(upr_problem) / sizeof(double));
    {
        const double* src = reinterpret_cast<const double*>(A.P);
#pragma unroll 4
        for (int i = threadIdx.x; i < NPD; i += NT) smem[i] = src[i];   // (four requests in flight per trip: 0.062 against 0.067 ms without)
        __syncthreads();
        A.P = reinterpret_cast<const upr_problem*>(smem);
    }
    upr_ls_instance<NQ, NFM, NBM, EXACT, OBS>(ctx, A, blockIdx.x, smem + ((NPD + 1) & ~1));
    if (A.xs_prev) {
        __syncthreads();
        const int b = blockIdx.x, nxs = (A.d.N + 1) * A.d.nx, nus = A.d.N * A.d.nu;
        const double* xs = A.xs + (size_t)b * nxs; const double* us = A.us + (size_t)b * nus;
        double* xp = A.xs_prev + (size_t)b * nxs; double* up = A.us_prev + (size_t)b * nus;
        for (int e = threadIdx.x; e < nxs; e += NT) xp[e] = xs[e];
        for (int e = threadIdx.x; e < nus; e += NT) up[e] = us[e];
        if (threadIdx.x == 0) A.tprev[b] = A.t0[b];
    }
}
#endif

// ---- warm start / policy evaluation ----------------------------------------------------------------
// Linear interpolation of a stored solution (ts = tp0 + j dt) at time tau; beyond the stored horizon
// the state is held and the input is zero (ocs2 DefaultInitializer, controller_interface.cpp:385-386).
static UPR_HDI void upr_interp(const upr_dims& d, double dt, double tp0, const double* xs, const double* us,
                                     double tau, int i_x, int i_u, double* xo, double* uo) {
    double s = (tau - tp0) / dt;
    int N = d.N;
    if (s < 0.0) s = 0.0;
    if (xo) {
        double v;
        if (s >= N) v = xs[N * d.nx + i_x];
        else { int j = (int)s; double a = s - j; v = (1.0 - a) * xs[j * d.nx + i_x] + a * xs[(j + 1) * d.nx + i_x]; }
        *xo = v;
    }
    if (uo) {
        double v;
        if (s > N) v = 0.0;
        else if (s >= N - 1) v = us[(N - 1) * d.nu + i_u];
        else { int j = (int)s; double a = s - j; v = (1.0 - a) * us[j * d.nu + i_u] + a * us[(j + 1) * d.nu + i_u]; }
        *uo = v;
    }
}
