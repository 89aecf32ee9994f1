// upr_common.h -- shared definitions of the HIP engine.
//
// Kernel bodies are written against three primitives -- UPR_FOR (block-strided loop), UPR_SYNC
// (workgroup barrier) and a per-workgroup scratch pointer -- so that the very same source also
// compiles with g++ as a one-thread-per-workgroup host emulation (-DUPR_HOST_EMU).  The emulation is a
// TEST-ONLY build (tests/emu/) for debugging index arithmetic without a GPU; libupright_mi.so never
// contains it and has no CPU path.
#pragma once
#include "../../include/upright_mi.h"

#ifdef UPR_HOST_EMU
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define UPR_HD
#define UPR_HDI inline
#define UPR_D
#define UPR_SYNC() ((void)0)
#define UPR_SYNC_LDS() ((void)0)
#define UPR_WSYNC() ((void)0)
#define UPR_WSYNC_LDS() ((void)0)
struct upr_ctx { int tid; int nt; };
#else
#include <hip/hip_runtime.h>
#define UPR_HD __host__ __device__
#define UPR_HDI __host__ __device__ __forceinline__
#define UPR_D __device__ __forceinline__
#define UPR_SYNC() __syncthreads()
// workgroup barrier that orders LDS traffic only: global loads in flight (register prefetches) and global
// stores that nobody reads before the next full UPR_SYNC are NOT waited for, unlike __syncthreads()
#define UPR_SYNC_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// wave-level ordering point: LDS operations of one wave are executed in program order, so only the
// compiler has to be kept from moving code across it
#define UPR_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
// the same for LDS traffic only: outstanding GLOBAL stores of the wave are not waited for (the fences above drain them)
#define UPR_WSYNC_LDS() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)
struct upr_ctx { int tid; int nt; };
#endif

// Block-strided loop.  On the device the start index passes through an empty asm: everything derived from it
// (addresses above all) is recomputed where the loop stands instead of being hoisted to the top of the
// kernel and kept alive -- spilled to scratch -- across the whole interior-point iteration.
#ifdef UPR_HOST_EMU
static inline int upr_opq(int x) { return x; }
#else
static __device__ __forceinline__ int upr_opq(int x) { asm volatile("" : "+v"(x)); return x; }
// quad permutation of a double by DPP (CTRL = quad_perm encoding: 0xB1 swaps neighbours, 0x4E swaps pairs); all
// four lanes of the quad must be active
template <int CTRL>
static __device__ __forceinline__ double upr_dpp_quad(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// value of lane `lane` (compile-time constant after unrolling) broadcast through scalar registers
static __device__ __forceinline__ double upr_readlane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// value of the lane whose BYTE address (4 * lane) is `addr4`, per lane (ds_bpermute_b32 twice)
static __device__ __forceinline__ double upr_bpermute(double v, int addr4) {
    const int lo = __builtin_amdgcn_ds_bpermute(addr4, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(addr4, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
#endif
#define UPR_FOR(i, n) for (int i = upr_opq(ctx.tid); i < (n); i += ctx.nt)

// 1/sqrt(x) without the IEEE division / square-root sequences (they cost ~300 cycles per pivot on the
// critical path): hardware estimate (relative error 5e-8 on gfx950, tools/rsq_test.hip) + ONE third-order
// step y (1 + e/2 + 3e^2/8), e = 1 - x y^2: 5 dependent operations, relative error < 3e-16.
static inline UPR_HD double upr_rsqrt(double x) {
#ifdef UPR_HOST_EMU
    return 1.0 / sqrt(x);
#else
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.0);
    return fma(y, e * fma(0.375, e, 0.5), y);
#endif
}

// 1/x the same way: hardware estimate + one second-order step y (1 + e + e^2), e = 1 - x y (the IEEE division
// sequence costs ~100 dependent cycles; the flat phases of the QP kernel are bound by two or three of them per row)
// z + a s as ONE fused operation, wherever the iterate takes a step: the rows' sweep forms the same new value ahead of the update
// of the iterate (upr_qp3.h, ineq_sweep what == 4) and both must round alike
static inline UPR_HD double upr_step(double z, double a, double s) { return __builtin_fma(a, s, z); }
static inline UPR_HD double upr_rcp(double x) {
#ifdef UPR_HOST_EMU
    return 1.0 / x;
#else
    const double y = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, y, 1.0);
    return fma(y, fma(e, e, e), y);
#endif
}

// derived dimensions
struct upr_dims {
    int nq, nb, nc, nf, N, nx, nu, nfc, ne, np, neN, no;   // no: collision pairs (state rows at knots 1..N-1)
    // per-knot linearisation record (doubles): [g ne][gx ne*nx][cost 1][grad nq][hess nq(nq+1)/2]
    int lin_g, lin_gx, lin_cost, lin_grad, lin_hess, lin_obs, lin_stride;   // lin_obs: [d no][dd/dq no*nq]
    // inequality layout per stage: [x lo nx][x hi nx] (k >= 1) [u lo nu][u hi nu][poly np] (k < N)
    int ni_stage;   // 2nx + 2nu + np  (slot size; stage 0 leaves the x part unused, stage N the u part)
    // Riccati stage store (doubles)
    int ss_kx, ss_hjj, ss_hff, ss_sinv, ss_ku0, ss_uf0, ss_snu, ss_pb, ss_stride;
    // per-instance workspace (doubles)
    int ws_dx, ws_du, ws_sx, ws_su, ws_pi, ws_nu, ws_yN, ws_pin, ws_nun, ws_dyN, ws_t, ws_lam, ws_rc, ws_store, ws_stride;
    int soft;                               // any soft row class (slack arrays below are empty otherwise)
    int ws_sig, ws_tau, ws_gam, ws_rcs;     // per slot: slack sigma, its own slack tau and multiplier gam, its complementarity target
    double eq_scale;                        // 1 / sqrt(6 nb): the scaling of the object-dynamics rows (balancing constraint's 1/sqrt(n))
    int n_dyn;                              // dynamic obstacles (their states are staged per knot by the linearisation kernel)
};

static inline UPR_HD upr_dims upr_make_dims(const upr_problem* P) {
    upr_dims d;
    d.nq = P->nq; d.nb = P->nb; d.nc = P->nc; d.nf = P->nf; d.N = P->N;
    d.nx = 3 * P->nq; d.nfc = P->nf * P->nc; d.nu = P->nq + d.nfc;
    d.ne = 6 * P->nb; d.np = (P->nf == 3) ? 5 * P->nc : 0;
    d.eq_scale = 1.0 / sqrt(6.0 * P->nb);
    d.neN = P->terminal_constraint ? 3 + 2 * P->nq : 0;
    d.lin_g = 0; d.lin_gx = d.ne; d.lin_cost = d.lin_gx + d.ne * d.nx; d.lin_grad = d.lin_cost + 1;
    d.no = P->n_pairs + P->n_proj; d.n_dyn = P->n_dyn;
    d.lin_hess = d.lin_grad + d.nq; d.lin_obs = d.lin_hess + d.nq * (d.nq + 1) / 2; d.lin_stride = d.lin_obs + d.no * (1 + d.nq);
    d.ni_stage = 2 * d.nx + 2 * d.nu + d.np + d.no;
    d.ss_kx = 0; d.ss_hjj = d.ss_kx + d.nq * d.nx; d.ss_hff = d.ss_hjj + d.nq * d.nq;
    d.ss_sinv = d.ss_hff + ((d.nf == 3) ? 9 * d.nc : d.nc);
    d.ss_ku0 = d.ss_sinv + d.ne * d.ne; d.ss_uf0 = d.ss_ku0 + d.nq; d.ss_snu = d.ss_uf0 + d.nfc;
    d.ss_pb = d.ss_snu + d.ne; d.ss_stride = d.ss_pb + d.nx;
    int n1 = d.N + 1;
    d.ws_dx = 0; d.ws_du = d.ws_dx + n1 * d.nx; d.ws_sx = d.ws_du + d.N * d.nu; d.ws_su = d.ws_sx + n1 * d.nx;
    d.ws_pi = d.ws_su + d.N * d.nu; d.ws_nu = d.ws_pi + n1 * d.nx; d.ws_yN = d.ws_nu + d.N * d.ne;
    d.ws_pin = d.ws_yN + d.neN; d.ws_nun = d.ws_pin + n1 * d.nx; d.ws_dyN = d.ws_nun + d.N * d.ne;
    d.ws_t = d.ws_dyN + d.neN; d.ws_lam = d.ws_t + n1 * d.ni_stage; d.ws_rc = d.ws_lam + n1 * d.ni_stage;
    d.soft = (P->soft_state_box || P->soft_input_box || P->soft_poly) ? 1 : 0;
    const int nsl = d.soft ? n1 * d.ni_stage : 0;
    d.ws_sig = d.ws_rc + n1 * d.ni_stage; d.ws_tau = d.ws_sig + nsl; d.ws_gam = d.ws_tau + nsl; d.ws_rcs = d.ws_gam + nsl;
    d.ws_store = d.ws_rcs + nsl;
    d.ws_stride = d.ws_store + d.N * d.ss_stride;
    d.ws_stride = (d.ws_stride + 15) & ~15;
    return d;
}

// packed upper-triangular index (i <= j) of an n x n symmetric matrix
static inline UPR_HD int upr_tri(int n, int i, int j) {
    if (i > j) { int t = i; i = j; j = t; }
    return i * n - (i * (i - 1)) / 2 + (j - i);
}

// reference_trajectory.h:18-47 (position part), ocs2 LinearInterpolation::timeSegment convention:
// value = alpha * lhs + (1 - alpha) * rhs
static inline UPR_HD void upr_target_position(const upr_problem* P, const double* way_p, double t, double* pd) {
    int n = P->n_way;
    if (n <= 1) { pd[0] = way_p[0]; pd[1] = way_p[1]; pd[2] = way_p[2]; return; }
    int idx; double alpha;
    if (t <= P->way_t[0]) { idx = 0; alpha = 1.0; }
    else if (t >= P->way_t[n - 1]) { idx = n - 2; alpha = 0.0; }
    else {
        idx = 0;
        while (idx + 1 < n - 1 && t >= P->way_t[idx + 1]) ++idx;
        alpha = (P->way_t[idx + 1] - t) / (P->way_t[idx + 1] - P->way_t[idx]);
    }
    for (int i = 0; i < 3; ++i) pd[i] = alpha * way_p[3 * idx + i] + (1.0 - alpha) * way_p[3 * (idx + 1) + i];
}

// reference_trajectory.h:18-47 (orientation part): the target orientation is the SLERP of the waypoint quaternions
// (xyzw), q_lhs.slerp(1 - alpha, q_rhs) with Eigen's rule (shortest arc; linear where the two nearly coincide); out: the
// rotation matrix (row-major) of the interpolated unit quaternion
static inline UPR_HD void upr_quat_to_rot(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
static inline UPR_HD void upr_target_rotation(const upr_problem* P, const double* way_q, double t, double* R) {
    const int n = P->n_way;
    if (n <= 1) { upr_quat_to_rot(way_q, R); return; }
    int idx; double alpha;
    if (t <= P->way_t[0]) { idx = 0; alpha = 1.0; }
    else if (t >= P->way_t[n - 1]) { idx = n - 2; alpha = 0.0; }
    else {
        idx = 0;
        while (idx + 1 < n - 1 && t >= P->way_t[idx + 1]) ++idx;
        alpha = (P->way_t[idx + 1] - t) / (P->way_t[idx + 1] - P->way_t[idx]);
    }
    const double* a = way_q + 4 * idx; const double* b = way_q + 4 * (idx + 1);
    const double s = 1.0 - alpha;
    const double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3], ad = fabs(d);
    double s0, s1;
    if (ad >= 1.0 - 2.220446049250313e-16) { s0 = 1.0 - s; s1 = s; }
    else { const double th = acos(ad), st = sin(th); s0 = sin((1.0 - s) * th) / st; s1 = sin(s * th) / st; }
    if (d < 0.0) s1 = -s1;
    double q[4];
    for (int i = 0; i < 4; ++i) q[i] = s0 * a[i] + s1 * b[i];
    const double nrm = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] *= nrm;
    upr_quat_to_rot(q, R);
}
static inline UPR_HD bool upr_has_orientation_cost(const upr_problem* P) { return P->Wee[3] != 0.0 || P->Wee[4] != 0.0 || P->Wee[5] != 0.0; }
