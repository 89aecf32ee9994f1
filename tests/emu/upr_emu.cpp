// upr_emu.cpp -- TEST-ONLY host emulation of the HIP kernel bodies (one thread per workgroup).
// Compiles upright_amd/csrc/*.h with -DUPR_HOST_EMU under g++ so that index arithmetic and the math
// of the kernels can be checked against the oracle without a GPU.  It is never part of
// libupright_mi.so and no product code path calls it.
#define UPR_HOST_EMU
#include <algorithm>
#include <vector>

#include "../../upright_amd/csrc/upr_common.h"
#include "../../upright_amd/csrc/upr_kin.h"
#include "../../upright_amd/csrc/upr_linearize.h"
#include "../../upright_amd/csrc/upr_linearize2.h"
#include "../../upright_amd/csrc/upr_linesearch.h"
#include "../../upright_amd/csrc/upr_qp.h"
#include "../../upright_amd/csrc/upr_qp2.h"
#include "../../upright_amd/csrc/upr_qp3.h"

template <int NQ, bool ORI>
static void lin_all_o(const upr_lin_args& A);
// (0: upr_linearize.h's phases for every shape; 1, the default: the job functions of upr_linearize2.h where the device runs them)
static int g_lin_form = 1;
template <int NQ>
static void lin_all(const upr_lin_args& A) {
    if (g_lin_form && upr_lin2_eligible(A)) {
        std::vector<double> sh(upr_lin2_layout(A.d, A.P->n_sph).per + 8);
        for (int p = 0; p < A.npoints; ++p) upr_lin2_knot<NQ>(A, upr_lin_locate(A, p), sh.data());
        return;
    }
    if (A.way_q) lin_all_o<NQ, true>(A); else lin_all_o<NQ, false>(A);
}
template <int NQ, bool ORI>
static void lin_all_o(const upr_lin_args& A) {
    std::vector<double> sh(upr_lin_lds_doubles(A.d, A.P->n_sph) + 8);
    for (int p = 0; p < A.npoints; ++p) {
        upr_lin_point q = upr_lin_locate(A, p);
        for (int l = 0; l < UPR_LPK; ++l) upr_lin_phase0(A, q, l, sh.data());
        if (UPR_LIN_SC_ONCE) for (int j = 0; j < NQ; ++j) upr_lin_phase0_sc(A, q, j, sh.data());
        for (int l = 0; l < UPR_LPK; ++l) upr_lin_phase1a<NQ>(A, q, l, sh.data());
        for (int l = 0; l < UPR_LPK; ++l) upr_lin_phase1<NQ, ORI>(A, q, l, sh.data());
        if (A.d.no > 0) {
#if UPR_LIN_OBS_SNAP && UPR_LIN_ANALYTIC
            if (A.dyn) for (int oi = 0; oi < A.P->n_dyn; ++oi) upr_lin_stage_obstacle(A, q, oi, sh.data());
#endif
            for (int l = 0; l < UPR_LPK; ++l) upr_lin_phase_obs_a<NQ>(A, q, l, sh.data());
            for (int l = 0; l < UPR_LPK; ++l) upr_lin_phase_obs_b<NQ>(A, q, l, sh.data());
        }
        for (int l = 0; l < UPR_LPK; ++l) upr_lin_phase2<NQ, ORI>(A, q, l, sh.data());
    }
}

template <class D>
static void qp2_all(const upr_qp_args& A, int B) {
    upr_ctx ctx; ctx.tid = 0; ctx.nt = 1;
    std::vector<double> L(upr_qp2_lds_doubles<D>(A.d.N, 1) + 16);
    for (int b = 0; b < B; ++b) upr_qp2_solve<D>(ctx, A, b, L.data());
}


template <int NQ>
static void ee_tangents(const upr_problem* P, const double* x, double* snap_form, double* walk_form) {
    double sc[2 * NQ], snap[NQ * UPR_SNAP_J + UPR_SNAP_E];
    for (int j = 0; j < NQ; ++j) upr_sincos(x[j], &sc[2 * j], &sc[2 * j + 1]);
    upr_ee_walk_snap<NQ>(P, x, sc, snap);
    auto dump = [](const upr_ee<upr_dd>& E, double* o) {
        int n = 0;
        for (int i = 0; i < 3; ++i) { o[n] = E.p[i].v; o[30 + n++] = E.p[i].d; }
        for (int i = 0; i < 9; ++i) { o[n] = E.C[i].v; o[30 + n++] = E.C[i].d; }
        for (int i = 0; i < 3; ++i) { o[n] = E.v[i].v; o[30 + n++] = E.v[i].d; }
        for (int i = 0; i < 3; ++i) { o[n] = E.w[i].v; o[30 + n++] = E.w[i].d; }
        for (int i = 0; i < 3; ++i) { o[n] = E.a[i].v; o[30 + n++] = E.a[i].d; }
        for (int i = 0; i < 3; ++i) { o[n] = E.al[i].v; o[30 + n++] = E.al[i].d; }
    };
    for (int dir = 0; dir < 3 * NQ; ++dir) {
        upr_ee<upr_dd> E;
        upr_ee_from_snap<NQ>(P, snap, dir, E); dump(E, snap_form + 60 * dir);
        upr_ee_kinematics<upr_dd, NQ>(P, x, dir, E, sc); dump(E, walk_form + 60 * dir);
    }
}

extern "C" {

void emu_set_lin_form(int form) { g_lin_form = form; }

// end-effector state [p 3, C 9, v 3, w 3, a 3, al 3] and its tangent along every state coordinate, [3 nq][2][30]: the closed
// form out of the per-joint snapshots (upr_ee_from_snap, what the linearisation kernel runs) and the forward-mode walk
void emu_ee_tangents(const upr_problem* P, const double* x, double* snap_form, double* walk_form) {
    if (P->nq == 6) ee_tangents<6>(P, x, snap_form, walk_form); else ee_tangents<9>(P, x, snap_form, walk_form);
}

void emu_dims(const upr_problem* P, int* out) {
    upr_dims d = upr_make_dims(P);
    out[0] = d.nx; out[1] = d.nu; out[2] = d.ne; out[3] = d.np; out[4] = d.lin_stride; out[5] = d.ws_stride;
    out[6] = d.ws_dx; out[7] = d.ws_du; out[8] = d.lin_g; out[9] = d.lin_gx; out[10] = d.lin_cost; out[11] = d.lin_grad; out[12] = d.lin_hess;
    out[13] = d.nfc;
}

// where the generic kernel leaves its multipliers in the instance workspace (tests/kkt_check.py)
void emu_kkt_offsets(const upr_problem* P, int* out) {
    upr_dims d = upr_make_dims(P);
    out[0] = d.ws_pi; out[1] = d.ws_nu; out[2] = d.ws_yN; out[3] = d.ws_lam; out[4] = d.ni_stage; out[5] = d.neN; out[6] = d.lin_obs; out[7] = d.no;
}

void emu_make_Df(const upr_problem* P, int B, const double* body_params, double* Df) {
    upr_dims d = upr_make_dims(P);
    std::vector<double> unit(d.nfc), Fw(6 * d.nb);
    const double scale = 1.0 / std::sqrt(6.0 * d.nb);
    for (int b = 0; b < B; ++b) {
        const double* bp = body_params + (size_t)b * d.nb * 10;
        for (int j = 0; j < d.nfc; ++j) {
            std::fill(unit.begin(), unit.end(), 0.0); unit[j] = 1.0;
            upr_object_wrenches(P, bp, unit.data(), Fw.data());
            for (int bb = 0; bb < d.nb; ++bb) for (int r = 0; r < 6; ++r)
                Df[((size_t)b * d.ne + 6 * bb + r) * d.nfc + j] = -scale * Fw[6 * bb + r] / bp[10 * bb];
        }
    }
}

// optional dynamic-obstacle data (test hook): observed state per instance and projectile flags
static const double* g_dyn = nullptr; static const double* g_pflag = nullptr;
void emu_set_dynamic(const double* dyn, const double* pflag) { g_dyn = dyn; g_pflag = pflag; }

static const double* g_way_q = nullptr;   // [B][n_way][4] target orientations (emu_set_way_q), NULL: none
void emu_set_way_q(const double* q) { g_way_q = q; }

void emu_linearize(const upr_problem* P, int B, const double* body_params, const double* way_p, const double* t0,
                   const double* xs, const double* us, double* lin) {
    upr_lin_args A;
    A.way_q = upr_has_orientation_cost(P) ? g_way_q : nullptr;
    A.P = P; A.d = upr_make_dims(P); A.body_params = body_params; A.way_p = way_p; A.t0 = t0; A.xs = xs; A.us = us;
    A.inst = nullptr; A.lin = lin; A.ee_out = nullptr; A.npoints = B * (P->N + 1);
    if (P->n_dyn) { A.dyn = g_dyn; A.pflag = g_pflag; }
    // the constant d g / d forces, as the engine hands it to the kernel (UPR_EMU_LIN_WRENCH=1: the wrench-sum form of the value)
    upr_dims d = A.d;
    std::vector<double> Df((size_t)B * d.ne * d.nfc);
    emu_make_Df(P, B, body_params, Df.data());
    if (!getenv("UPR_EMU_LIN_WRENCH")) A.Df = Df.data();
    if (P->nq == 6) lin_all<6>(A); else lin_all<9>(A);
}

void emu_qp(const upr_problem* P, int B, const double* xs, const double* us, const double* x0, const double* lin,
            const double* Df, double* ws, double* stats) {
    upr_qp_args A;
    A.P = P; A.d = upr_make_dims(P); A.xs = xs; A.us = us; A.x0 = x0; A.lin = lin; A.Df = Df; A.ws = ws; A.stats = stats; A.prof = nullptr;
    upr_ctx ctx; ctx.tid = 0; ctx.nt = 1;
    // LDS is not zero on the device: poison the scratch so that a read before the first write cannot pass unnoticed
    std::vector<double> L(upr_qp_lds_layout(A.d, 1).total + 16, std::nan(""));
    for (int b = 0; b < B; ++b) upr_qp_solve(ctx, A, b, L.data());
}

// third-structure kernel body (NT = 1 emulation of the <9,1,4,3,N=20> instantiation)
long emu_qp3(const upr_problem* P, int B, const double* xs, const double* us, const double* x0, const double* lin,
             const double* Df, double* ws, long ws_stride, double* stats) {
    upr_qp_args A;
    A.P = P; A.d = upr_make_dims(P); A.xs = xs; A.us = us; A.x0 = x0; A.lin = lin; A.Df = Df; A.ws = ws; A.stats = stats; A.prof = nullptr;
    const bool softb = P->soft_state_box || P->soft_input_box || (P->soft_poly && (A.d.np > 0 || A.d.no > 0));   // upr_api.hip needs_soft
    // the instantiations libupright_mi launches (upr_api.hip: headline, UPR_QP3_EXTRA), one thread per workgroup
#define EMU_QP3(a, b, c, e, sf, cond) EMU_QP3N(a, b, c, e, 20, sf, false, cond)
#define EMU_QP3D(a, b, c, e, sf, dense, cond) EMU_QP3N(a, b, c, e, 20, sf, dense, cond)
#define EMU_QP3N(a, b, c, e, n, sf, dense, cond) if (P->nq == a && P->nb == b && P->nc == c && P->nf == e && P->N == n && (cond)) { \
        typedef upr_qp3_cfg<a, b, c, e, n, 1, true, sf, dense> C; \
        if (!ws) return (long)upr_qp3_ws<C>::total; \
        A.d.ws_stride = (int)ws_stride; \
        upr_ctx ctx; ctx.tid = 0; ctx.nt = 1; \
        std::vector<double> L(upr_qp3_lds<C>::total + 16, std::nan("")); \
        for (int bb = 0; bb < B; ++bb) upr_qp3_solve<C>(ctx, A, bb, L.data()); \
        return 0; }
    EMU_QP3(9, 1, 4, 3, false, !softb)
    EMU_QP3(9, 1, 4, 3, true, softb)
    EMU_QP3(9, 1, 4, 1, true, true)
    EMU_QP3(9, 8, 32, 1, true, true)
    EMU_QP3D(9, 3, 16, 3, false, true, !softb)
    EMU_QP3N(6, 1, 4, 1, 20, true, false, true)
    EMU_QP3N(6, 1, 4, 1, 10, true, false, true)
    EMU_QP3D(9, 7, 28, 3, false, false, !softb)   // round 4: seven cups (star, friction: BIGF), two stacked dice, arm-only with friction
    EMU_QP3D(9, 2, 8, 3, false, true, !softb)
    EMU_QP3N(6, 1, 4, 3, 20, false, false, !softb)
#undef EMU_QP3
#undef EMU_QP3D
#undef EMU_QP3N
    return -1;
}
long emu_qp3_lds_doubles() { return (long)upr_qp3_lds<upr_qp3_cfg<9, 1, 4, 3, 20, 256>>::total; }

// production kernel body; returns the per-instance workspace size it needs (doubles) when ws == NULL
long emu_qp2(const upr_problem* P, int B, const double* xs, const double* us, const double* x0, const double* lin,
             const double* Df, double* ws, long ws_stride, double* stats) {
    upr_qp_args A;
    A.P = P; A.d = upr_make_dims(P); A.xs = xs; A.us = us; A.x0 = x0; A.lin = lin; A.Df = Df; A.ws = ws; A.stats = stats; A.prof = nullptr;
    if (P->nq == 9 && P->nb == 1 && P->nc == 4 && P->nf == 3) {
        typedef upr_qp2_dims<9, 1, 4, 3> D;
        if (!ws) return (long)upr_qp2_ws_doubles<D>(A.d.N, A.d.neN);
        A.d.ws_stride = (int)ws_stride;
        qp2_all<D>(A, B);
        return 0;
    }
    return -1;
}

void emu_linesearch(const upr_problem* P, int B, double* xs, double* us, const double* x0, const double* t0,
                    const double* body_params, const double* way_p, const double* lin, const double* ws, long ws_stride, double* stats,
                    int* done, int iter) {
    upr_ls_args A;
    A.P = P; A.d = upr_make_dims(P); if (ws_stride > 0) A.d.ws_stride = (int)ws_stride; A.xs = xs; A.us = us; A.x0 = x0; A.t0 = t0; A.body_params = body_params; A.way_p = way_p;
    A.lin = lin; A.ws = ws; A.stats = stats; A.done = done; A.iter = iter; A.way_q = upr_has_orientation_cost(P) ? g_way_q : nullptr;
    if (P->n_dyn) { A.dyn = g_dyn; A.pflag = g_pflag; }
    upr_ls_lanes ctx; ctx.tid = 0; ctx.nt = 1; ctx.ftid = 0; ctx.fnt = 1;
    std::vector<double> L(upr_ls_lds_doubles(A.d) + 16);
    bool acc = false;
    for (int b = 0; b < B; ++b) { if (P->nq == 6) upr_ls_instance<6>(ctx, A, b, L.data(), A.done[b], false, false, A.stats[(size_t)b * UPR_NSTATS + 2], &acc, A.way_p + (size_t)b * P->n_way * 3, A.t0[b]); else upr_ls_instance<9>(ctx, A, b, L.data(), A.done[b], false, false, A.stats[(size_t)b * UPR_NSTATS + 2], &acc, A.way_p + (size_t)b * P->n_way * 3, A.t0[b]); }
}
}
