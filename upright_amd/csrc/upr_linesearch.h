// upr_linesearch.h -- merit evaluation, filter line search and SQP convergence test, one workgroup (two waves) per
// instance: a lane per shooting knot for the terms that need the chain walk, a lane per element for the rest.
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
    int n_way = 0;         // P->n_way (the device kernel stages the instance's waypoints before its copy of the record is readable)
};

// lanes of a workgroup as the line search uses them: [tid, nt) strides the per-knot work (the chain walks, a lane per knot: the
// first lanes, i.e. wave 0 for the horizons in use), [ftid, fnt) strides the FLAT work (everything that needs no walk, a lane per
// element) -- on the device the lanes behind wave 0, so that the flat sums of a trial run beside its walks (ftid < 0: none);
// the host emulation is one lane doing both
struct upr_ls_lanes { int tid, nt, ftid, fnt; };

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

// Performance terms [cost, dyn_sse, eq_sse, ineq_sse] of a trajectory (Xs [N+1][nx], Us [N][nu]), in two parts.
//
// (1) upr_ls_knot: what needs the end effector's state at the knot -- end-effector cost, object-dynamics equality, collision /
//     projectile rows, terminal position error -- a lane per knot walking the chain on plain values.
// (2) upr_ls_flat_terms: everything else -- quadratic state-input cost, boxes, friction rows, dynamics defects, terminal
//     velocity / acceleration -- element by element over the flat lanes (a lane per knot summing its 27 + 21 elements one
//     after another was half of the kernel's run time: one wave, 21 of its lanes, ~500 LDS reads each).
//     BASE: at the CURRENT iterate, where the linearisation kernel has just been, part (1) is read out of the knots' records
//     instead of walking again, and the step norms and the Armijo descent metric (cost gradient . step) are summed in the same pass.
//
// NFM / NBM: compile-time bounds of nf nc / nb.  EXACT: the problem has exactly nf = 3, nc = NFM / 3 contacts, nb = NBM bodies
// (checked by the launcher; OBS: with collision / projectile rows): every loop bound of the knot's part is then a constant and its
// input vector and wrenches stay in registers (with run-time bounds they were indexed dynamically: 1.2 KB of scratch per lane).
// Xt / Ut: the trial trajectory of the instance (staged by the workgroup, LDS on the device); sc: (sin, cos) of the trial joint
// angles of every knot, [N + 1][NQ][2]; pd: the knot's target position (it does not depend on the step length)
template <int NQ, int NFM = 3 * UPR_MAX_CONTACTS, int NBM = UPR_MAX_BODIES, bool EXACT = false, bool OBS = !EXACT>
static UPR_HDI void upr_ls_knot(const upr_ls_args& A, int b, int k, const double* Xt, const double* Ut, const double* sc, const double* pd, double* out) {
    const upr_problem* P = A.P; const upr_dims& d = A.d;
    const int N = d.N;
    constexpr int nq = NQ, nx = 3 * NQ;
    const int nu = EXACT ? NQ + NFM : d.nu;
    const double h = P->dt;
    const double* X = Xt + k * nx;   // (read where it lies: the rolled joint loop indexes it by the joint)
    double F[NFM];
    double cost = 0.0, eq = 0.0, iq = 0.0;
    // (the body's parameters and the wrench in front of the chain walk: their global loads fly beside it)
    double Fw[6 * NBM];
    const int nb = EXACT ? NBM : d.nb;
    const double* bp = A.body_params + (size_t)b * nb * 10;
    double bpv[EXACT ? 10 * NBM : 1];
    if (k < N) {
#pragma unroll
        for (int i = 0; i < (EXACT ? NFM : d.nfc); ++i) F[i] = Ut[k * nu + nq + i];
        if (EXACT) {
#pragma unroll
            for (int i = 0; i < 10 * NBM; ++i) bpv[i] = bp[i];
        }
        if (EXACT && NBM == 1) upr_object_wrench_single<NFM / 3>(P, bpv, F, Fw);
        else upr_object_wrenches(P, bp, F, Fw);
    }
    if (OBS && d.no > 0 && k >= 1 && k < N) {   // collision rows (knots 1..N-1); !OBS: the problem has none
        double dd[UPR_MAX_PAIRS + 8], xo[9 * UPR_MAX_DYN];
        if (A.dyn) for (int oi = 0; oi < P->n_dyn; ++oi) upr_obstacle_at(A.dyn + ((size_t)b * P->n_dyn + oi) * 9, k * h, xo + 9 * oi, xo + 9 * oi + 3, xo + 9 * oi + 6);
        upr_obstacle_values<NQ>(P, X, A.dyn ? xo : nullptr, A.pflag ? A.pflag[b] : 0.0, dd);
        for (int r = 0; r < d.no; ++r) { const double v = fmin(0.0, dd[r]); iq += h * v * v; }
    }
    upr_ee<double> E;
    upr_ee_kinematics<double, NQ, true>(P, X, -1, E, sc + k * 2 * NQ);
    if (k < N) {
        double c = 0.0;
        for (int r = 0; r < 3; ++r) { double e = E.p[r] - pd[r]; c += 0.5 * P->Wee[r] * e * e; }
        if (A.way_q) {
            double Rr[9], eo[3];
            upr_target_rotation(P, A.way_q + (size_t)b * P->n_way * 4, A.t0[b] + k * h, Rr);
            upr_orientation_error<double>(E.C, Rr, eo);
            for (int r = 0; r < 3; ++r) c += 0.5 * P->Wee[3 + r] * eo[r] * eo[r];
        }
        cost += h * c;
        // object-dynamics equality
        const double scl = A.d.eq_scale;
#pragma unroll
        for (int bb = 0; bb < (EXACT ? NBM : nb); ++bb) {
            double g[6];
            upr_body_residual<double>(E, (EXACT ? bpv : bp) + 10 * bb, P->gravity, Fw + 6 * bb, Fw + 6 * bb + 3, g);
            for (int r = 0; r < 6; ++r) eq += h * (scl * g[r]) * (scl * g[r]);
        }
    } else if (d.neN > 0) {
        for (int r = 0; r < 3; ++r) { double e = pd[r] - E.p[r]; eq += e * e; }
    }
    out[0] += cost; out[2] += eq; out[3] += iq;
}

// aux (BASE only) += [descent metric, |dx|^2, |du|^2, -].  NU / NC / NE: nu, nc, ne where the instantiation fixes them (0: run
// time; the element -> (knot, index) split is then an integer division by a register, ~40 instructions a trip).
// BASE reads the knots' records and x0 from global memory: the first UPR_LS_RU trips of each such sum are REQUESTED in front of
// the LDS-only sums and consumed behind them (inside the strided loops every trip waited for its own request: five exposed
// memory latencies for the headline's states alone); longer sums finish in plain loops.
#define UPR_LS_RU 2
struct upr_ls_rec { double gq[UPR_LS_RU], g[UPR_LS_RU], c, x0, t; };
// (clamped addresses: every lane requests, the out-of-range ones are not used)
template <int NQ, int NE>
static UPR_HDI void upr_ls_rec_request(const upr_ls_args& A, int b, int ftid, int fnt, upr_ls_rec& R) {
    const upr_dims& d = A.d;
    const int N = d.N, ne = NE ? NE : d.ne;
    constexpr int nq = NQ, nx = 3 * NQ;
    const double* lin = A.lin + (size_t)b * (N + 1) * d.lin_stride;
#pragma unroll
    for (int u = 0; u < UPR_LS_RU; ++u) {
        const int e = u * fnt + ftid, e1 = (e < N * nq) ? e : 0, k1 = e1 / nq, e2 = (e < N * ne) ? e : 0, k2 = e2 / ne;
        R.gq[u] = lin[(size_t)k1 * d.lin_stride + d.lin_grad + (e1 - k1 * nq)];
        R.g[u] = lin[(size_t)k2 * d.lin_stride + d.lin_g + (e2 - k2 * ne)];
    }
    R.c = lin[(size_t)((ftid < N) ? ftid : 0) * d.lin_stride + d.lin_cost];
    R.x0 = A.x0[(size_t)b * nx + ((ftid < nx) ? ftid : 0)];
    R.t = lin[(size_t)N * d.lin_stride + d.lin_grad + ((ftid < 3) ? ftid : 0)];
}
template <int NQ, bool BASE, int NU = 0, int NC = 0, int NE = 0>
static UPR_HDI void upr_ls_flat_terms(const upr_ls_args& A, int b, int ftid, int fnt, const double* Xs, const double* Us, const double* dxs, const double* dus,
                                      double* out, double* aux) {
    if (ftid < 0) return;
    const upr_problem* P = A.P; const upr_dims& d = A.d;
    const int N = d.N, nu = NU ? NU : d.nu, nc = NC ? NC : d.nc, ne = NE ? NE : d.ne;
    constexpr int nq = NQ, nx = 3 * NQ;
    const double h = P->dt, h2 = 0.5 * h * h, h3 = h * h * h / 6.0;
    const int nxs = (N + 1) * nx, nus = N * nu;
    const double* lin = A.lin + (size_t)b * (N + 1) * d.lin_stride;
    double cost = 0.0, dyn = 0.0, eq = 0.0, iq = 0.0, a0 = 0.0, a1 = 0.0, a2 = 0.0;
    upr_ls_rec R;
    if (BASE) upr_ls_rec_request<NQ, NE>(A, b, ftid, fnt, R);
    // states: box, quadratic cost, terminal velocity / acceleration (+ BASE: the state part of the descent metric, |dx|^2)
    for (int e = ftid; e < nxs; e += fnt) {
        const int k = e / nx, i = e - k * nx;
        const double X = Xs[e];
        if (k >= 1) { const double v = fmin(0.0, fmin(X - P->x_lb[i], P->x_ub[i] - X)); iq += ((k < N) ? h : 1.0) * v * v; }
        if (k < N) {
            const double ex = X - P->xd[i];
            cost += h * (0.5 * P->Qdiag[i] * ex * ex);
            if (BASE) a0 += P->dt * (P->Qdiag[i] * ex) * dxs[e];
        } else if (d.neN > 0 && i >= nq) eq += X * X;
        if (BASE) { const double s = dxs[e]; a1 += s * s; }
    }
    // inputs: quadratic cost, box
    for (int e = ftid; e < nus; e += fnt) {
        const int k = e / nu, i = e - k * nu;
        const double U = Us[e];
        cost += h * (0.5 * P->Rdiag[i] * U * U);
        const double v = fmin(0.0, fmin(U - P->u_lb[i], P->u_ub[i] - U));
        iq += h * v * v;
        if (BASE) { const double s = dus[e]; a2 += s * s; a0 += P->dt * P->Rdiag[i] * U * s; }
    }
    // dynamics defect against the next state (triple integrator, system_dynamics.h:15-22)
    for (int e = ftid; e < N * nq; e += fnt) {
        const int k = e / nq, j = e - k * nq;
        const double* x = Xs + k * nx; const double* xn = x + nx;
        const double q = x[j], v = x[nq + j], a = x[2 * nq + j], u = Us[k * nu + j];
        const double e0 = q + h * v + h2 * a + h3 * u - xn[j], e1 = v + h * a + h2 * u - xn[nq + j], e2 = a + h * u - xn[2 * nq + j];
        dyn += h * (e0 * e0 + e1 * e1 + e2 * e2);
    }
    // friction rows
    if (d.np > 0) for (int e = ftid; e < N * nc; e += fnt) {
        const int k = e / nc, ci = e - k * nc;
        double hr[5];
        upr_friction_rows_contact(P, ci, Us + k * nu + nq + 3 * ci, hr);
        for (int r = 0; r < 5; ++r) { const double v = fmin(0.0, hr[r]); iq += h * v * v; }
    }
    if (!BASE) {   // initial-state defect
        for (int i = ftid; i < nx; i += fnt) { const double ee = A.x0[(size_t)b * nx + i] - Xs[i]; dyn += ee * ee; }
    } else {
        // the knots' part out of the records of the linearisation kernel (end-effector cost and its gradient, equality rows,
        // terminal position error, collision rows), the initial-state defect
#pragma unroll
        for (int u = 0; u < UPR_LS_RU; ++u) {
            const int e = u * fnt + ftid;
            if (e < N * nq) { const int k = e / nq, i = e - k * nq; a0 += P->dt * R.gq[u] * dxs[k * nx + i]; }
            if (e < N * ne) eq += h * R.g[u] * R.g[u];
        }
        for (int e = UPR_LS_RU * fnt + ftid; e < N * nq; e += fnt) { const int k = e / nq, i = e - k * nq; a0 += P->dt * lin[(size_t)k * d.lin_stride + d.lin_grad + i] * dxs[k * nx + i]; }
        for (int e = UPR_LS_RU * fnt + ftid; e < N * ne; e += fnt) { const int k = e / ne, r = e - k * ne; const double g = lin[(size_t)k * d.lin_stride + d.lin_g + r]; eq += h * g * g; }
        if (ftid < N) cost += h * R.c;
        for (int k = fnt + ftid; k < N; k += fnt) cost += h * lin[(size_t)k * d.lin_stride + d.lin_cost];
        if (ftid < nx) { const double ee = R.x0 - Xs[ftid]; dyn += ee * ee; }
        for (int i = fnt + ftid; i < nx; i += fnt) { const double ee = A.x0[(size_t)b * nx + i] - Xs[i]; dyn += ee * ee; }
        if (d.neN > 0 && ftid < 3) eq += R.t * R.t;
        if (d.neN > 0) for (int r = fnt + ftid; r < 3; r += fnt) { const double g = lin[(size_t)N * d.lin_stride + d.lin_grad + r]; eq += g * g; }
        if (d.no > 0) for (int e = ftid; e < (N - 1) * d.no; e += fnt) {
            const int k = 1 + e / d.no, r = e - (k - 1) * d.no;
            const double v = fmin(0.0, lin[(size_t)k * d.lin_stride + d.lin_obs + r]); iq += h * v * v;
        }
        aux[0] += a0; aux[1] += a1; aux[2] += a2;
    }
    out[0] += cost; out[1] += dyn; out[2] += eq; out[3] += iq;
}

// Workgroup scratch (doubles).  [0, UPR_LS_RED): reduction slots; then the trial trajectory xs + alpha dx, us + alpha du of the
// step length in work, (sin, cos) of its joint angles (a lane per (knot, joint): inside the chain walk they were nine f64
// sincos in series per lane), the knots' target positions, and (stage_full) the instance's trajectory and step, staged with
// coalesced requests (stage_full = 0 -- long horizons of the large shapes, where three copies do not fit: the trajectory and
// the step are read where they lie)
#define UPR_LS_RED 72   // 8 sums x 4 waves, twice (alternating), + the ranks of the waves
struct upr_ls_lay { int xt, ut, sc, pd, wp, sx, su, sdx, sdu, total; };
static UPR_HDI upr_ls_lay upr_ls_layout(const upr_dims& d, bool full) {
    upr_ls_lay l;
    const int nxs = (d.N + 1) * d.nx, nus = d.N * d.nu;
    l.xt = UPR_LS_RED; l.ut = l.xt + nxs; l.sc = l.ut + nus; l.pd = l.sc + 2 * (d.N + 1) * d.nq;
    l.wp = l.pd + 3 * (d.N + 1);   // the instance's waypoints
    l.sx = (l.wp + 3 * UPR_MAX_WAYPOINTS + 1) & ~1; l.su = l.sx + nxs; l.sdx = l.su + nus; l.sdu = l.sdx + nxs;
    l.total = (full ? l.sdu + nus : l.sx) + 2;
    return l;
}
static UPR_HDI int upr_ls_lds_doubles(const upr_dims& d, bool full = true) { return upr_ls_layout(d, full).total; }

#if defined(UPR_LS_PROF) && !defined(UPR_HOST_EMU)
// instrumented build (-DUPR_LS_PROF): cycles from the start of the kernel to each stamp, summed over the workgroups by lane 0
// ([15] = workgroups counted; upr_debug_ls_prof reads them)
__device__ unsigned long long upr_ls_prof[16];
#define UPR_LS_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const long long now_ = __builtin_readcyclecounter(); if (ctx.tid == 0) atomicAdd(&upr_ls_prof[i], (unsigned long long)(now_ - t_prof)); __builtin_amdgcn_sched_barrier(0); } while (0)
#define UPR_LS_PROF_ARG , long long t_prof
#define UPR_LS_PROF_PASS , t_prof
#else
#define UPR_LS_STAMP(i) ((void)0)
#define UPR_LS_PROF_ARG
#define UPR_LS_PROF_PASS
#endif

// workgroup sum of NV partials, every lane ends with the sums.  Device: butterflies inside the wave, one LDS slot per wave and
// sum, ONE barrier (the callers alternate the slot set `par`, so the next reduction's writes cannot overtake this one's reads)
template <int NV>
static UPR_HDI void upr_ls_reduce(const upr_ls_lanes& ctx, double* L, int par, const double* part, double* res) {
#ifndef UPR_HOST_EMU
    const int nw = ctx.nt >> 6, w = ctx.tid >> 6;
    double* slot = L + par * 32;
#pragma unroll
    for (int c = 0; c < NV; ++c) {   // (cross-lane moves inside the VALU: quads, half rows, rows, then the four rows through scalar registers)
        double v = part[c];
        v += upr_dpp_quad<0xB1>(v); v += upr_dpp_quad<0x4E>(v); v += upr_dpp_quad<0x141>(v); v += upr_dpp_quad<0x140>(v);
        v = (upr_readlane(v, 0) + upr_readlane(v, 16)) + (upr_readlane(v, 32) + upr_readlane(v, 48));
        if ((ctx.tid & 63) == 0) slot[c * 4 + w] = v;
    }
    UPR_SYNC();
#pragma unroll
    for (int c = 0; c < NV; ++c) { double v = slot[c * 4]; for (int i = 1; i < nw; ++i) v += slot[c * 4 + i]; res[c] = v; }
#else
    for (int c = 0; c < NV; ++c) res[c] = part[c];
#endif
}

// done_b: the instance's convergence flag and qp_status: the status of its QP (stats[2]); staged: the caller has already
// staged xs, us, dx, du into the layout (the device kernel requests all of these beside its first loads); defer_store: an
// accepted trajectory is written to xs, us by the caller (the device kernel: one pass with the remembered solution); way_p_b,
// t0_b: the instance's waypoints (the device kernel's copy in LDS) and its time.  Returns the copy of the instance's (possibly updated) trajectory in LDS,
// [xs (N+1) nx, us N nu], or NULL when there is none (instance done; stage_full off and the step rejected): the kernel's
// epilogue copies the remembered solution from it.
// STAGE = A.stage_full, at compile time on the device: the staged arrays are then LDS pointers for the compiler (chosen at run
// time they were generic pointers -- flat loads, each waiting for every global request in flight as well)
template <int NQ, int NFM = 3 * UPR_MAX_CONTACTS, int NBM = UPR_MAX_BODIES, bool EXACT = false, bool OBS = !EXACT, int STAGE = -1>
static UPR_HDI const double* upr_ls_instance(const upr_ls_lanes& ctx, const upr_ls_args& A, int b, double* L, int done_b, bool staged, bool defer_store, double qp_status, bool* accepted_out,
                                             const double* way_p_b, double t0_b UPR_LS_PROF_ARG) {
    const upr_problem* P = A.P; const upr_dims& d = A.d;
    const int N = d.N, nx = d.nx, nu = d.nu, nq = d.nq;
    if (done_b) return nullptr;
    double* st = A.stats + (size_t)b * UPR_NSTATS;
    UPR_LS_STAMP(1);
    const double alpha_decay = 0.5, alpha_min = 1e-4, gamma_c = 1e-6, g_max = 1e6, g_min = 1e-6, armijo = 1e-4;
    double* xs = A.xs + (size_t)b * (N + 1) * nx; double* us = A.us + (size_t)b * N * nu;
    const double* ws = A.ws + (size_t)b * d.ws_stride;
    const double* dx = ws + d.ws_dx; const double* du = ws + d.ws_du;
    const int nxs = (N + 1) * nx, nus = N * nu;
    const bool stage_full = (STAGE < 0) ? (A.stage_full != 0) : (STAGE != 0);
    const upr_ls_lay lay = upr_ls_layout(d, stage_full);
    double* Xt = L + lay.xt; double* Ut = L + lay.ut; double* sc = L + lay.sc; double* pdl = L + lay.pd;
    const double* xs_l = xs; const double* us_l = us; const double* dx_l = dx; const double* du_l = du;
    if (stage_full) {
        double* sx_ = L + lay.sx; double* su_ = L + lay.su; double* sdx_ = L + lay.sdx; double* sdu_ = L + lay.sdu;
        if (!staged) {
            UPR_FOR(i, nxs) { sx_[i] = xs[i]; sdx_[i] = dx[i]; }
            UPR_FOR(i, nus) { su_[i] = us[i]; sdu_[i] = du[i]; }
            UPR_SYNC();
        }
        xs_l = sx_; us_l = su_; dx_l = sdx_; du_l = sdu_;
    }
    UPR_LS_STAMP(2);
    // baseline, step norms and Armijo descent metric: one flat pass over all the lanes, one reduction
    double part[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bs[8];
    upr_ls_flat_terms<NQ, true, EXACT ? NQ + NFM : 0, EXACT ? NFM / 3 : 0, EXACT ? 6 * NBM : 0>(A, b, ctx.tid, ctx.nt, xs_l, us_l, dx_l, du_l, part, part + 4);
    // the knots' target positions (a lane's own knots: written and read by the same lane; the waypoint reads fly beside the
    // reduction and the first trial's staging)
    if (qp_status != 2.0) { UPR_FOR(k, N + 1) upr_target_position(P, way_p_b, t0_b + k * P->dt, pdl + 3 * k); }
    UPR_LS_STAMP(3);
    upr_ls_reduce<7>(ctx, L, 0, part, bs);
    UPR_LS_STAMP(4);
    const double base[4] = {bs[0], bs[1], bs[2], bs[3]};
    const double descent = bs[4], dxn = sqrt(bs[5]), dun = sqrt(bs[6]);
    const double base_viol = sqrt(base[1] + base[2] + base[3]);
    double alpha = 1.0, perf[4] = {base[0], base[1], base[2], base[3]};
    bool accepted = false;
    if (qp_status != 2.0) {
        int par = 1;
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
            UPR_LS_STAMP(5);
            UPR_FOR(k, N + 1) upr_ls_knot<NQ, NFM, NBM, EXACT, OBS>(A, b, k, Xt, Ut, sc, pdl + 3 * k, p2);
            upr_ls_flat_terms<NQ, false, EXACT ? NQ + NFM : 0, EXACT ? NFM / 3 : 0, EXACT ? 6 * NBM : 0>(A, b, ctx.ftid, ctx.fnt, Xt, Ut, nullptr, nullptr, p2, nullptr);
            UPR_LS_STAMP(6);
            upr_ls_reduce<4>(ctx, L, par, p2, perf);
            par ^= 1;
            double viol = sqrt(perf[1] + perf[2] + perf[3]);
            if (viol > g_max) accepted = false;
            else if (viol < g_min) accepted = (descent < 0.0) ? (perf[0] < base[0] + armijo * alpha * descent) : true;
            else accepted = (perf[0] < base[0] - gamma_c * base_viol) || (viol < (1.0 - gamma_c) * base_viol);
            if (accepted) break;
            alpha *= alpha_decay;
            UPR_SYNC();   // (the next trial overwrites Xt, Ut, sc)
        } while (alpha >= alpha_min);
    }
    UPR_LS_STAMP(7);
    double cost = base[0], viol = base_viol;
    if (accepted) {
        if (!defer_store) {
            UPR_FOR(i, nxs) xs[i] = Xt[i];   // (the trial trajectory of the accepted step length: xs + alpha dx, as staged)
            UPR_FOR(i, nus) us[i] = Ut[i];
        }
        cost = perf[0]; viol = sqrt(perf[1] + perf[2] + perf[3]);
    }
    *accepted_out = accepted;
    bool conv = !accepted;                                                           // STEPSIZE
    if (accepted && fabs(base[0] - cost) < P->cost_tol && viol < g_min) conv = true;  // METRICS
    if (accepted && alpha * dxn < P->delta_tol && alpha * dun < P->delta_tol) conv = true;  // PRIMAL
    if (ctx.tid == 0) {
        st[0] = A.iter + 1; st[3] = accepted ? alpha : 0.0; st[4] = cost; st[5] = viol; st[10] = dxn; st[11] = dun;
        if (conv) A.done[b] = 1;
    }
    UPR_LS_STAMP(8);
    return accepted ? Xt : (stage_full ? xs_l : nullptr);   // (Xt, Ut contiguous; so are the staged xs, us)
}

#ifndef UPR_HOST_EMU
// One workgroup of NT = 256 lanes per instance.  Wave 0 carries the chain walks of a trial (a lane per knot), the other waves
// its flat sums; the prologue (copy of the problem record, dispatch rank, staging) has every global request of the workgroup in
// flight before the first wait.
template <int NQ, int NT, int NFM = 3 * UPR_MAX_CONTACTS, int NBM = UPR_MAX_BODIES, bool EXACT = false, bool OBS = !EXACT, bool STAGE = true>
__global__ void __launch_bounds__(NT, NT / 64) upr_linesearch_kernel(upr_ls_args A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    static_assert(NT % 64 == 0 && NT >= 128 && NT <= 256, "two to four waves: the rank and the reductions keep four slots; wave 0 walks");
    upr_ls_lanes ctx; ctx.tid = threadIdx.x; ctx.nt = NT; ctx.ftid = (int)threadIdx.x - 64; ctx.fnt = NT - 64;
#ifdef UPR_LS_PROF
    const long long t_prof = __builtin_readcyclecounter();
    if (threadIdx.x == 0) atomicAdd(&upr_ls_prof[15], 1ull);
#endif
    // The problem record (joint frames, bounds, weights, contacts: 11 KB) is read all over the evaluation, element by element
    // and mostly on the serial chain walk -- out of global memory every one of those reads is a vector load with a wait of
    // its own (the record may alias the kernel's stores, so they are not scalar loads): a copy in LDS serves them instead.
    static_assert(sizeof(upr_problem) % sizeof(double) == 0, "copied as doubles");
    constexpr int NPD = (int)(sizeof(upr_problem) / sizeof(double)), NTR = (NPD + NT - 1) / NT;
    double* L = smem + ((NPD + 1) & ~1);
    const int b = blockIdx.x;
    const upr_dims& d = A.d;
    const int nxs = (d.N + 1) * d.nx, nus = d.N * d.nu;
    double recv[NTR];
    {
        const double* src = reinterpret_cast<const double*>(A.P);
#pragma unroll
        for (int t = 0; t < NTR; ++t) { const int i = t * NT + threadIdx.x; recv[t] = src[(i < NPD) ? i : 0]; }
    }
    const int done_b = A.done[b];
    const double qp_status_b = A.stats[(size_t)b * UPR_NSTATS + 2], t0_b = A.t0[b];
    const double wpv = (A.n_way > 0) ? A.way_p[(size_t)b * A.n_way * 3 + ((int)threadIdx.x < 3 * A.n_way ? threadIdx.x : 0)] : 0.0;   // (3 n_way <= NT)
    // the instance's trajectory and step, requested beside the record (up to SU elements of each per lane; longer ones are
    // staged by upr_ls_instance)
    constexpr int SU = 1024 / NT;
    const bool stage_here = STAGE && nxs <= SU * NT && nus <= SU * NT;
    const upr_ls_lay lay = upr_ls_layout(d, STAGE);
    double sxv[SU], sdxv[SU], suv[SU], sduv[SU];
    if (stage_here) {
        const double* xs = A.xs + (size_t)b * nxs; const double* us = A.us + (size_t)b * nus;
        const double* dx = A.ws + (size_t)b * d.ws_stride + d.ws_dx; const double* du = A.ws + (size_t)b * d.ws_stride + d.ws_du;
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int i = u * NT + threadIdx.x, ix = (i < nxs) ? i : 0, iu = (i < nus) ? i : 0;
            sxv[u] = xs[ix]; sdxv[u] = dx[ix]; suv[u] = us[iu]; sduv[u] = du[iu];
        }
    }
    if (A.order_out) {
        // B one-byte keys, read four at a time by consecutive lanes (B bytes per workgroup out of the L2: 16 MB per launch at
        // B = 4096 where ranking on stats[.][1], one 96-byte stride per instance, moved 1 GB)
        const int B = gridDim.x, me = b;
        const int mk = A.iter_key[me];
        int cnt = 0;
        const unsigned int* k4 = reinterpret_cast<const unsigned int*>(A.iter_key);
        for (int o0 = 0; o0 < (B >> 2); o0 += 4 * NT) {   // (four requests per lane in flight)
            unsigned int w4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int o4 = o0 + u * NT + threadIdx.x; w4[u] = k4[(o4 < (B >> 2)) ? o4 : 0]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int o4 = o0 + u * NT + threadIdx.x;
                if (o4 < (B >> 2)) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const int k = (int)((w4[u] >> (8 * j)) & 255u), o = 4 * o4 + j; cnt += (k > mk || (k == mk && o < me)) ? 1 : 0; }
                }
            }
        }
        for (int o = (B & ~3) + threadIdx.x; o < B; o += NT) { const int k = A.iter_key[o]; cnt += (k > mk || (k == mk && o < me)) ? 1 : 0; }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
        if ((threadIdx.x & 63) == 0) reinterpret_cast<int*>(L + 64)[threadIdx.x >> 6] = cnt;   // (summed behind the barrier below)
    }
    {
#pragma unroll
        for (int t = 0; t < NTR; ++t) { const int i = t * NT + threadIdx.x; if (i < NPD) smem[i] = recv[t]; }
        if ((int)threadIdx.x < 3 * A.n_way) L[lay.wp + threadIdx.x] = wpv;
        if (stage_here) {
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int i = u * NT + threadIdx.x;
                if (i < nxs) { L[lay.sx + i] = sxv[u]; L[lay.sdx + i] = sdxv[u]; }
                if (i < nus) { L[lay.su + i] = suv[u]; L[lay.sdu + i] = sduv[u]; }
            }
        }
        __syncthreads();
        A.P = reinterpret_cast<const upr_problem*>(smem);
    }
    if (A.order_out && threadIdx.x == 0) { const int* c = reinterpret_cast<const int*>(L + 64); int r = 0; for (int w = 0; w < NT / 64; ++w) r += c[w]; A.order_out[r] = b; }
    UPR_LS_STAMP(0);
    bool accepted = false;
    const double* kept = upr_ls_instance<NQ, NFM, NBM, EXACT, OBS, STAGE ? 1 : 0>(ctx, A, b, L, done_b, stage_here, true, qp_status_b, &accepted, L + lay.wp, t0_b UPR_LS_PROF_PASS);
    UPR_LS_STAMP(9);
    if (accepted) {   // the accepted trajectory: one pass over its copy in LDS for the iterate and the remembered solution
        double* xs = A.xs + (size_t)b * nxs; double* us = A.us + (size_t)b * nus;
        double* xp = A.xs_prev ? A.xs_prev + (size_t)b * nxs : nullptr; double* up = A.xs_prev ? A.us_prev + (size_t)b * nus : nullptr;
#pragma unroll 4
        for (int e = threadIdx.x; e < nxs; e += NT) { const double v = kept[e]; xs[e] = v; if (xp) xp[e] = v; }
#pragma unroll 4
        for (int e = threadIdx.x; e < nus; e += NT) { const double v = kept[nxs + e]; us[e] = v; if (up) up[e] = v; }
        if (A.xs_prev && threadIdx.x == 0) A.tprev[b] = A.t0[b];
    } else if (A.xs_prev) {
        double* xp = A.xs_prev + (size_t)b * nxs; double* up = A.us_prev + (size_t)b * nus;
        if (kept) {   // the trajectory as the line search left it lies in LDS: stores only, no round trip through memory
            for (int e = threadIdx.x; e < nxs; e += NT) xp[e] = kept[e];
            for (int e = threadIdx.x; e < nus; e += NT) up[e] = kept[nxs + e];
        } else {
            const double* xs = A.xs + (size_t)b * nxs; const double* us = A.us + (size_t)b * nus;
            for (int e = threadIdx.x; e < nxs; e += NT) xp[e] = xs[e];
            for (int e = threadIdx.x; e < nus; e += NT) up[e] = us[e];
        }
        if (threadIdx.x == 0) A.tprev[b] = A.t0[b];
    }
    UPR_LS_STAMP(10);
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
