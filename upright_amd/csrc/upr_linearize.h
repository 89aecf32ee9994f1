// upr_linearize.h -- per-knot linearisation of the OCP terms (kernels K1..K5 of SURVEY.md 2b, fused).
//
// One group of LPK = 32 lanes per shooting knot (two knots per 64-wide wavefront).  Lane l < nx
// carries the forward-mode tangent along state coordinate l through
//   end-effector kinematics  ->  object-dynamics residual (balancing_constraints.cpp:114-155)
// and writes column l of d g / d x; the value lane writes g.  The end-effector cost
// (cost/end_effector_cost.h:48-84) gets its gradient J'We and Gauss-Newton Hessian J'WJ from the
// position tangents exchanged through LDS (VALU path) or through one v_mfma_f64_16x16x4_f64 per knot
// (MFMA path).  Friction-cone rows (contact_constraints.h:50-77) are linear in u with a constant
// Jacobian, so they are formed where they are consumed (QP / line-search kernels), not stored.
//
// Record per knot (doubles), see upr_dims: [g ne][gx ne*nx][cost][grad nq][hess nq(nq+1)/2].
// Terminal knot (k == N, stationary_desired_position_constraint.h:39-74): grad[0..2] = p_d - p,
// hess[0..3nq) = J_p (3 x nq row-major); g/gx unused.
//
// Algorithmic HBM bytes per knot: 8 * [(nx + nu) + ne (1 + nx) + (nq + nq(nq+1)/2 + 1)].
#pragma once
#include "upr_kin.h"

#define UPR_LPK 32
#ifndef UPR_LIN_ANALYTIC
#define UPR_LIN_ANALYTIC 1   // 0: every tangent lane walks the chain itself on (value, tangent) pairs (rounds 1 - 2; A/B runs)
#endif

// LDS per knot (doubles): x[nx] u[nu] Fw[6 nb] J[6 nq] e[6] Rref[9] sincos[2 nq] | sphere centres and their q-tangents [ns][3][1 + nq]
// (J / e: position rows, then the orientation rows used when the end-effector cost weighs orientation)
static UPR_HDI int upr_lin_lds_sc(const upr_dims& d) { return d.nx + d.nu + 6 * d.nb + 6 * d.nq + 6 + 9; }   // (sin q_j, cos q_j) [nq][2]
static UPR_HDI int upr_lin_lds_base(const upr_dims& d) { return (upr_lin_lds_sc(d) + 2 * d.nq + 1 + 1) & ~1; }
// [sphere centres ...] then the snapshots of the one value walk per knot (upr_kin.h: UPR_SNAP_J per joint + UPR_SNAP_E)
// UPR_LIN_OBS_SNAP = 1 (round 4): the collision rows out of the value walk's snapshots -- the area holds the sphere centres
// [ns][3] and the link frames [nq][UPR_SNAP_F] the walk leaves for placing them; 0: rounds 2 - 3, nq forward-mode walks of the
// spheres per knot, centres and their q-tangents [ns][3][1 + nq] (A/B builds)
#ifndef UPR_LIN_OBS_SNAP
#define UPR_LIN_OBS_SNAP 1
#endif
static UPR_HDI int upr_lin_lds_frames(const upr_dims& d, int n_sph) { return upr_lin_lds_base(d) + ((n_sph * 3 + 1) & ~1); }
// (+ the dynamic obstacles' states at the knot, [UPR_MAX_DYN][r 3, v 3, a 3]: staged at the top of the kernel, so that no row
// waits for a global load of its own)
static UPR_HDI int upr_lin_lds_dyn(const upr_dims& d, int n_sph) { return upr_lin_lds_frames(d, n_sph) + d.nq * UPR_SNAP_F; }
static UPR_HDI int upr_lin_lds_snap(const upr_dims& d, int n_sph = 0) {
#if UPR_LIN_OBS_SNAP && UPR_LIN_ANALYTIC
    // (the obstacle area only where there are obstacles: 37 doubles per knot put configs[2]'s sixteen knots beyond 80 KB, i.e. one
    //  workgroup per CU instead of two: 0.39 -> 0.57 ms)
    return d.no > 0 ? upr_lin_lds_dyn(d, n_sph) + (d.n_dyn > 0 ? ((9 * d.n_dyn + 1 + 1) & ~1) : 0) : upr_lin_lds_base(d);
#else
    return upr_lin_lds_base(d) + (d.no > 0 ? n_sph * 3 * (1 + d.nq) : 0);
#endif
}
static UPR_HDI int upr_lin_lds_doubles(const upr_dims& d, int n_sph = 0) { return (upr_lin_lds_snap(d, n_sph) + d.nq * UPR_SNAP_J + UPR_SNAP_E + 1) & ~1; }

struct upr_lin_args {
    const upr_problem* P;
    upr_dims d;
    const double* body_params;  // [B][nb][10]
    const double* way_p;        // [B][n_way][3]
    const double* way_q = nullptr;   // [B][n_way][4] target orientations (xyzw); NULL unless Wee[3..5] != 0 (set by the host: the
                                     // kernel never looks the weights up to decide)
    const double* t0;           // [B] (trajectory mode) or per point (points mode)
    const double* xs;           // trajectory mode: [B][N+1][nx]; points mode: [n][nx]
    const double* us;           // trajectory mode: [B][N][nu];   points mode: [n][nu]
    const int* inst;            // points mode: instance of each point; NULL in trajectory mode
    double* lin;                // [npoints][lin_stride]
    double* ee_out;             // optional [npoints][3] end-effector position
    int npoints;
    // dynamic obstacle: observed state [r, v, a] per instance (trajectory mode: propagated to the knot) or per point
    // (points mode, taken as is); activation flag of the projectile rows per instance.  NULL when n_dyn == 0.
    const double* dyn = nullptr;
    const double* pflag = nullptr;
    // optional: the constant d g / d forces of every instance, [B][ne][nfc] (the QP's Df).  The equality is affine in the
    // forces, g = g(x; f = 0) + Df f, so with it the force part of the VALUE is ne short dot products spread over the
    // lanes of the knot instead of one lane summing contact wrenches (a chain of dependent loads and 6 divisions per contact)
    const double* Df = nullptr;
};

struct upr_lin_point {
    int p, b, k;
    bool terminal;
    double t;
    const double* x;
    const double* u;
    double* out;
};

static UPR_HDI upr_lin_point upr_lin_locate(const upr_lin_args& A, int p) {
    upr_lin_point q;
    const upr_dims& d = A.d;
    q.p = p;
    if (A.inst) {
        q.b = A.inst[p]; q.k = 0; q.terminal = false; q.t = A.t0[p];
        q.x = A.xs + (size_t)p * d.nx; q.u = A.us + (size_t)p * d.nu;
    } else {
        q.b = p / (d.N + 1); q.k = p % (d.N + 1); q.terminal = (q.k == d.N);
        q.t = A.t0[q.b] + q.k * A.P->dt;
        q.x = A.xs + (size_t)p * d.nx;
        int ku = q.terminal ? d.N - 1 : q.k;
        q.u = A.us + ((size_t)q.b * d.N + ku) * d.nu;
    }
    q.out = A.lin + (size_t)p * d.lin_stride;
    return q;
}

// phase 0: stage x, u into LDS (coalesced: lane l loads element l) -- lanes 0..LPK-1 of the knot.  Two of the lanes that
// carry no tangent work from global memory at the same time: the summed contact wrench per body (its forces straight from
// the input vector, so it does not wait for the staging) and the desired orientation of the knot (SLERP of the waypoints)
// sin / cos of joint angle j of the knot, straight from the input (no wait for the staging).  The f64 sincos is a few hundred
// instructions whichever lanes run it: UPR_LIN_SC_ONCE = 1 runs it ONCE per workgroup, a lane per (knot, joint) over all the
// passes' knots (rounds 1 - 3a: once per pass, on nq of the knot's 32 lanes)
#ifndef UPR_LIN_SC_ONCE
#define UPR_LIN_SC_ONCE 1
#endif
static UPR_HDI void upr_lin_phase0_sc(const upr_lin_args& A, const upr_lin_point& q, int j, double* sh) {
    double s_, c_;
    upr_sincos(q.x[j], &s_, &c_);
    sh[upr_lin_lds_sc(A.d) + 2 * j] = s_; sh[upr_lin_lds_sc(A.d) + 2 * j + 1] = c_;
}
static UPR_HDI void upr_lin_phase0(const upr_lin_args& A, const upr_lin_point& q, int lane, double* sh) {
    const upr_dims& d = A.d;
    double* sx = sh; double* su = sh + d.nx;
    for (int i = lane; i < d.nx; i += UPR_LPK) sx[i] = q.x[i];
    for (int i = lane; i < d.nu; i += UPR_LPK) su[i] = q.terminal ? 0.0 : q.u[i];
#if !UPR_LIN_SC_ONCE
    // sin / cos of the joint angles once per knot (lanes nq .. 2 nq - 1, straight from the input: no wait for the staging)
    if (lane >= d.nq && lane < 2 * d.nq) upr_lin_phase0_sc(A, q, lane - d.nq, sh);
#endif
    if (A.Df != nullptr) {
        if (!q.terminal) {
            double* gf = sh + d.nx + d.nu;   // Df f of this knot, in the slot of the wrenches (6 nb doubles)
            const double* f = q.u + d.nq;
            for (int r = lane; r < d.ne; r += UPR_LPK) {
                const double* D = A.Df + ((size_t)q.b * d.ne + r) * d.nfc;
                double v = 0.0;
                int j = 0;
                for (; j + 4 <= d.nfc; j += 4) {   // four operands of each kind requested together
                    const double d0 = D[j], d1 = D[j + 1], d2 = D[j + 2], d3 = D[j + 3], f0 = f[j], f1 = f[j + 1], f2 = f[j + 2], f3 = f[j + 3];
                    v += d0 * f0; v += d1 * f1; v += d2 * f2; v += d3 * f3;
                }
                for (; j < d.nfc; ++j) v += D[j] * f[j];
                gf[r] = v;
            }
        }
    } else if (lane == UPR_LPK - 1 && !q.terminal) {
        double* Fw = sh + d.nx + d.nu;
        const double* bp = A.body_params + (size_t)q.b * d.nb * 10;
        if (d.nc <= 4) upr_object_wrenches_small<4>(A.P, bp, q.u + d.nq, Fw);
        else upr_object_wrenches(A.P, bp, q.u + d.nq, Fw);
    }
    if (lane == UPR_LPK - 2 && A.way_q != nullptr)
        upr_target_rotation(A.P, A.way_q + (size_t)q.b * A.P->n_way * 4, q.t, sh + d.nx + d.nu + 6 * d.nb + 6 * d.nq + 6);
}

#ifndef UPR_HOST_EMU
// Device form of phase 0 over ALL the passes' knots of the workgroup at once (the per-lane form above, pass by pass, keeps one
// or two requests in flight per lane and pays a memory round trip per pass and per dependent step): every lane first REQUESTS
// its elements of x and u of all passes, then the rows of Df f are spread a lane per (knot, row) with the row of Df and the
// forces of the knot requested together, then everything is stored.  Same arithmetic, same order of the sums.
#ifndef UPR_LIN_P0_BATCH
#define UPR_LIN_P0_BATCH 1
#endif
#ifndef UPR_LIN_OVERLAP
#define UPR_LIN_OVERLAP 1   // the value walks (wave 0) run beside the staging (waves 1 - 3)
#endif
// G0: the first of the eight 32-lane groups that takes part (G0 = 2: wave 0 walks the chains meanwhile, see the kernel)
template <int NQ, int NP, int G0 = 0>
static __device__ __forceinline__ void upr_lin_phase0_batched(const upr_lin_args& A, int base, int per, double* smem) {
    const upr_dims& d = A.d;
    constexpr int NX = 3 * NQ, NG = 8 - G0, NR = (8 * NP + NG - 1) / NG;
    static_assert(NX <= UPR_LPK, "one state element per lane");
    const int sub = (threadIdx.x >> 5) - G0, lane = threadIdx.x & 31;
    double xv[NR], uv[NR][2];
#pragma unroll
    for (int pp = 0; pp < NR; ++pp) {
        const int p = base + pp * NG + sub;
        xv[pp] = 0.0; uv[pp][0] = 0.0; uv[pp][1] = 0.0;
        if (pp * NG + sub < 8 * NP && p < A.npoints) {
            const upr_lin_point q = upr_lin_locate(A, p);
            if (lane < NX) xv[pp] = q.x[lane];
            if (!q.terminal) {
                if (lane < d.nu) uv[pp][0] = q.u[lane];
                if (lane + UPR_LPK < d.nu) uv[pp][1] = q.u[lane + UPR_LPK];
            }
        }
    }
    if (A.Df != nullptr) {
        const int tot = 8 * NP * d.ne;
        for (int idx = threadIdx.x - 32 * G0; idx < tot; idx += 32 * NG) {
            const int sp = idx / d.ne, r = idx - sp * d.ne, p = base + sp;
            if (p >= A.npoints) continue;
            const upr_lin_point q = upr_lin_locate(A, p);
            if (q.terminal) continue;
            const double* f = q.u + d.nq;
            const double* D = A.Df + ((size_t)q.b * d.ne + r) * d.nfc;
            double v = 0.0;
            if (d.nfc == 12) {
                double dv[12], fv[12];
#pragma unroll
                for (int j = 0; j < 12; ++j) { dv[j] = D[j]; fv[j] = f[j]; }
#pragma unroll
                for (int j = 0; j < 12; ++j) v += dv[j] * fv[j];
            } else {
                int j = 0;
                for (; j + 4 <= d.nfc; j += 4) {
                    const double d0 = D[j], d1 = D[j + 1], d2 = D[j + 2], d3 = D[j + 3], f0 = f[j], f1 = f[j + 1], f2 = f[j + 2], f3 = f[j + 3];
                    v += d0 * f0; v += d1 * f1; v += d2 * f2; v += d3 * f3;
                }
                for (; j < d.nfc; ++j) v += D[j] * f[j];
            }
            smem[sp * per + d.nx + d.nu + r] = v;
        }
    }
#pragma unroll
    for (int pp = 0; pp < NR; ++pp) {
        const int slot = pp * NG + sub, p = base + slot;
        if (slot < 8 * NP && p < A.npoints) {
            double* sh = smem + slot * per;
            if (lane < NX) sh[lane] = xv[pp];
            if (lane < d.nu) sh[NX + lane] = uv[pp][0];
            if (lane + UPR_LPK < d.nu) sh[NX + lane + UPR_LPK] = uv[pp][1];
            const upr_lin_point q = upr_lin_locate(A, p);
            for (int i = lane + 2 * UPR_LPK; i < d.nu; i += UPR_LPK) sh[NX + i] = q.terminal ? 0.0 : q.u[i];
            if (A.Df == nullptr && lane == UPR_LPK - 1 && !q.terminal) {
                double* Fw = sh + d.nx + d.nu;
                const double* bp = A.body_params + (size_t)q.b * d.nb * 10;
                if (d.nc <= 4) upr_object_wrenches_small<4>(A.P, bp, q.u + d.nq, Fw);
                else upr_object_wrenches(A.P, bp, q.u + d.nq, Fw);
            }
            if (lane == UPR_LPK - 2 && A.way_q != nullptr)
                upr_target_rotation(A.P, A.way_q + (size_t)q.b * A.P->n_way * 4, q.t, sh + d.nx + d.nu + 6 * d.nb + 6 * d.nq + 6);
        }
    }
}
#endif

// phase 1a (UPR_LIN_ANALYTIC): ONE walk of the chain per knot on plain values, by the knot's first lane; the tangent lanes
// of phase 1 read the snapshot of their joint (upr_kin.h, "analytic tangents")
template <int NQ>
static UPR_HDI void upr_lin_phase1a(const upr_lin_args& A, const upr_lin_point& q, int lane, double* sh, const double* x = nullptr) {
#if UPR_LIN_ANALYTIC
    // (x: the state straight from the input when the staging into LDS runs at the same time on other waves)
    if (lane == 0) upr_ee_walk_snap<NQ>(A.P, x ? x : sh, sh + upr_lin_lds_sc(A.d), sh + upr_lin_lds_snap(A.d, A.P->n_sph),
                                        (UPR_LIN_OBS_SNAP && A.d.no > 0) ? sh + upr_lin_lds_frames(A.d, A.P->n_sph) : nullptr);
#endif
}

// phase 1: the tangent lanes.  ORI: the end-effector cost weighs orientation (compile-time: the error-row loops keep
// constant trip counts)
template <int NQ, bool ORI = false>
static UPR_HDI void upr_lin_phase1(const upr_lin_args& A, const upr_lin_point& q, int lane, double* sh) {
    const upr_dims& d = A.d;
    const upr_problem* P = A.P;
    const double* sx = sh;
    const double* Fw = sh + d.nx + d.nu;
    double* sJ = sh + d.nx + d.nu + 6 * d.nb;
    double* se = sJ + 6 * d.nq;
    constexpr bool ori = ORI;
    const int dir = (lane < d.nx) ? lane : -1;
    upr_ee<upr_dd> E;
#if UPR_LIN_ANALYTIC
    upr_ee_from_snap<NQ>(P, sh + upr_lin_lds_snap(d, P->n_sph), dir, E);
#else
    upr_ee_kinematics<upr_dd, NQ>(P, sx, dir, E, sh + upr_lin_lds_sc(d));
#endif
    if (!q.terminal) {
        const double scale = d.eq_scale;   // (1 / sqrt(6 nb), from the host: an f64 sqrt and a division per lane and pass otherwise)
        const double* bp = A.body_params + (size_t)q.b * d.nb * 10;
        const double zero3[3] = {0.0, 0.0, 0.0};
        for (int b = 0; b < d.nb; ++b) {
            upr_dd gb[6];
            if (A.Df != nullptr) upr_body_residual<upr_dd>(E, bp + 10 * b, P->gravity, zero3, zero3, gb);
            else upr_body_residual<upr_dd>(E, bp + 10 * b, P->gravity, Fw + 6 * b, Fw + 6 * b + 3, gb);
            for (int r = 0; r < 6; ++r) {
                if (dir >= 0) q.out[d.lin_gx + (6 * b + r) * d.nx + dir] = scale * gb[r].d;
                if (lane == 0) q.out[d.lin_g + 6 * b + r] = scale * gb[r].v + (A.Df != nullptr ? Fw[6 * b + r] : 0.0);
            }
        }
    }
    if (dir >= 0 && dir < NQ)
        for (int r = 0; r < 3; ++r) sJ[r * NQ + dir] = E.p[r].d;
    if (ori && !q.terminal) {
        upr_dd eo[3];
        upr_orientation_error<upr_dd>(E.C, se + 6, eo);
        if (dir >= 0 && dir < NQ) for (int r = 0; r < 3; ++r) sJ[(3 + r) * NQ + dir] = eo[r].d;
        if (lane == 0) for (int r = 0; r < 3; ++r) se[3 + r] = eo[r].v;
    }
    if (lane == 0) {
        double pd[3];
        upr_target_position(P, A.way_p + (size_t)q.b * P->n_way * 3, q.t, pd);
        for (int r = 0; r < 3; ++r) se[r] = E.p[r].v - pd[r];
        if (A.ee_out) for (int r = 0; r < 3; ++r) A.ee_out[(size_t)q.p * 3 + r] = E.p[r].v;
    }
}

// collision rows (only when the problem has pairs), two sub-phases around a barrier:
//   a: lane l < nq walks the chain with the tangent along q_l and leaves every sphere centre's tangent in LDS
//      (lane 0 also the values);  b: lane r owns pair r: distance and its gradient n . (dc_a/dq - dc_b/dq)
// obstacle state at this point (trajectory mode: k dt after the observation)
// (obstacle oi of the point's instance; A.dyn is [instance | point][n_dyn][9])
static UPR_HDI void upr_lin_obstacle(const upr_lin_args& A, const upr_lin_point& q, int oi, double* ro, double* vo, double* ao) {
    for (int i = 0; i < 3; ++i) { ro[i] = 0.0; vo[i] = 0.0; ao[i] = 0.0; }
    if (!A.dyn) return;
    const int nd = A.P->n_dyn;
    if (A.inst) upr_obstacle_at(A.dyn + ((size_t)q.p * nd + oi) * 9, 0.0, ro, vo, ao);
    else upr_obstacle_at(A.dyn + ((size_t)q.b * nd + oi) * 9, q.k * A.P->dt, ro, vo, ao);
}
#if UPR_LIN_OBS_SNAP && UPR_LIN_ANALYTIC
// Snapshot form (round 4).  The value walk of phase 1a left, per joint j, the origin o_j and the axis z_j (snapshots) and the
// link frame behind the joint (frames).  a: lane s owns sphere s and places its centre from the frame it rides on (world,
// link j, tool, or obstacle i) -- no walk;  b: lane r owns row r: value and unit direction n as before, and the gradient in
// closed form: a sphere on link f moves rigidly with every joint j <= f, so
//     d c / d q_j = z_j x (c - o_j)   (revolute),   z_j   (prismatic),   0   (j > f, world and obstacle spheres),
// i.e. (d row / d q_j) = w n . (t_a,j - t_b,j).  The lanes of a knot sit in one wave: a wave-local ordering point separates
// the two (no workgroup barrier), and the per-knot LDS area shrinks from [ns][3][1 + nq] to [ns][3] + [nq][12].
// state of obstacle oi at the knot into the knot's LDS area (any lane; before the barrier in front of the rows)
static UPR_HDI void upr_lin_stage_obstacle(const upr_lin_args& A, const upr_lin_point& q, int oi, double* sh) {
    double* D = sh + upr_lin_lds_dyn(A.d, A.P->n_sph) + 9 * oi;
    upr_lin_obstacle(A, q, oi, D, D + 3, D + 6);
    if (oi == 0) sh[upr_lin_lds_dyn(A.d, A.P->n_sph) + 9 * A.d.n_dyn] = A.pflag ? A.pflag[q.b] : 0.0;   // activation flag of the projectile rows
}
// one sphere of one knot
template <int NQ>
static UPR_HDI void upr_lin_obs_sphere(const upr_lin_args& A, const upr_lin_point& q, int s, double* sh) {
    const upr_problem* P = A.P;
    double* sc = sh + upr_lin_lds_base(A.d);
    const double* fr = sh + upr_lin_lds_frames(A.d, P->n_sph);
    const double* T = sh + upr_lin_lds_snap(A.d, P->n_sph) + NQ * UPR_SNAP_J;
    if (q.terminal) return;
    {
        const int f = P->sph_frame[s];
        const double* off = P->sph_off[s];
        double c[3];
        if (f <= -2) {
            const double* ro = sh + upr_lin_lds_dyn(A.d, P->n_sph) + 9 * (-2 - f);
            for (int i = 0; i < 3; ++i) c[i] = ro[i] + off[i];
        } else if (f < 0) {
            for (int i = 0; i < 3; ++i) c[i] = off[i];
        } else {
            const double* Cf = (f >= NQ) ? T : fr + f * UPR_SNAP_F;
            const double* pf = (f >= NQ) ? T + 9 : Cf + 9;
            for (int i = 0; i < 3; ++i) c[i] = pf[i] + Cf[3 * i] * off[0] + Cf[3 * i + 1] * off[1] + Cf[3 * i + 2] * off[2];
        }
        for (int i = 0; i < 3; ++i) sc[3 * s + i] = c[i];
    }
}
template <int NQ>
static UPR_HDI void upr_lin_phase_obs_a(const upr_lin_args& A, const upr_lin_point& q, int lane, double* sh) {
    for (int s = lane; s < A.P->n_sph; s += UPR_LPK) upr_lin_obs_sphere<NQ>(A, q, s, sh);
}
// one row of one knot
template <int NQ>
static UPR_HDI void upr_lin_obs_row(const upr_lin_args& A, const upr_lin_point& q, int r, const double* sh) {
    const upr_problem* P = A.P; const upr_dims& d = A.d;
    const double* sc = sh + upr_lin_lds_base(d);
    const double* snap = sh + upr_lin_lds_snap(d, P->n_sph);
    if (q.terminal) return;
    // (the projectile rows follow the last obstacle: state.tail(9); its state at the knot was staged at the top of the kernel)
    const double* ro = sh + upr_lin_lds_dyn(d, P->n_sph) + 9 * (P->n_dyn > 0 ? P->n_dyn - 1 : 0);
    const double* vo = ro + 3; const double* ao = ro + 6;
    const double flag = (A.dyn && r >= P->n_pairs) ? sh[upr_lin_lds_dyn(d, P->n_sph) + 9 * d.n_dyn] : 0.0;
    {
        int sa, sb; double n[3], w;
        q.out[d.lin_obs + r] = upr_state_row(P, r, [&](int s, int i) { return sc[3 * s + i]; }, ro, vo, ao, flag, &sa, &sb, n, &w);
        const int fa = P->sph_frame[sa], fb = (sb >= 0) ? P->sph_frame[sb] : -1;
        double ca[3], cb[3];
        for (int i = 0; i < 3; ++i) { ca[i] = sc[3 * sa + i]; cb[i] = (sb >= 0) ? sc[3 * sb + i] : 0.0; }
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const double* S = snap + j * UPR_SNAP_J;
            const double o[3] = {S[0], S[1], S[2]}, z[3] = {S[15], S[16], S[17]};
            const bool rev = P->joint_type[j] == 1;
            // n . (z x (c - o)) = (c - o) . (n x z): the two spheres differ only in c
            double nz[3];
            upr_cross(n, z, nz);
            double v = 0.0;
            if (rev) {
                if (fa >= j) v += (ca[0] - o[0]) * nz[0] + (ca[1] - o[1]) * nz[1] + (ca[2] - o[2]) * nz[2];
                if (fb >= j) v -= (cb[0] - o[0]) * nz[0] + (cb[1] - o[1]) * nz[1] + (cb[2] - o[2]) * nz[2];
            } else {
                const double nd = n[0] * z[0] + n[1] * z[1] + n[2] * z[2];
                if (fa >= j) v += nd;
                if (fb >= j) v -= nd;
            }
            q.out[d.lin_obs + d.no + r * NQ + j] = w * v;
        }
    }
}
template <int NQ>
static UPR_HDI void upr_lin_phase_obs_b(const upr_lin_args& A, const upr_lin_point& q, int lane, const double* sh) {
    for (int r = lane; r < A.d.no; r += UPR_LPK) upr_lin_obs_row<NQ>(A, q, r, sh);
}
#else
template <int NQ>
static UPR_HDI void upr_lin_phase_obs_a(const upr_lin_args& A, const upr_lin_point& q, int lane, double* sh) {
    const upr_problem* P = A.P;
    double* sc = sh + upr_lin_lds_base(A.d);
    if (q.terminal || lane >= NQ) return;
    for (int s = 0; s < P->n_sph; ++s) if (P->sph_frame[s] <= -2) {   // rides on an obstacle: no dependence on q
        double ro[3], vo[3], ao[3];
        upr_lin_obstacle(A, q, -2 - P->sph_frame[s], ro, vo, ao);
        for (int i = 0; i < 3; ++i) { sc[(s * 3 + i) * (1 + NQ) + 1 + lane] = 0.0; if (lane == 0) sc[(s * 3 + i) * (1 + NQ)] = ro[i] + P->sph_off[s][i]; }
    }
    upr_sphere_walk<upr_dd, NQ>(P, sh, lane, [&](int s, const upr_dd* c) {
        for (int i = 0; i < 3; ++i) {
            sc[(s * 3 + i) * (1 + NQ) + 1 + lane] = c[i].d;
            if (lane == 0) sc[(s * 3 + i) * (1 + NQ)] = c[i].v;
        }
    });
}
template <int NQ>
static UPR_HDI void upr_lin_phase_obs_b(const upr_lin_args& A, const upr_lin_point& q, int lane, const double* sh) {
    const upr_problem* P = A.P; const upr_dims& d = A.d;
    const double* sc = sh + upr_lin_lds_base(d);
    if (q.terminal) return;
    double ro[3], vo[3], ao[3];
    upr_lin_obstacle(A, q, A.P->n_dyn > 0 ? A.P->n_dyn - 1 : 0, ro, vo, ao);   // (the projectile rows follow the last obstacle: state.tail(9))
    const double flag = A.pflag ? A.pflag[q.b] : 0.0;
    for (int r = lane; r < d.no; r += UPR_LPK) {
        int sa, sb; double n[3], w;
        q.out[d.lin_obs + r] = upr_state_row(P, r, [&](int s, int i) { return sc[(s * 3 + i) * (1 + NQ)]; }, ro, vo, ao, flag, &sa, &sb, n, &w);
        for (int j = 0; j < NQ; ++j) {
            double v = 0.0;
            for (int i = 0; i < 3; ++i) v += n[i] * (sc[(sa * 3 + i) * (1 + NQ) + 1 + j] - (sb >= 0 ? sc[(sb * 3 + i) * (1 + NQ) + 1 + j] : 0.0));
            q.out[d.lin_obs + d.no + r * NQ + j] = w * v;
        }
    }
}
#endif

// phase 2 (VALU path): gradient, Gauss-Newton Hessian, cost from the LDS-staged position Jacobian
template <int NQ, bool ORI = false>
static UPR_HDI void upr_lin_phase2(const upr_lin_args& A, const upr_lin_point& q, int lane, const double* sh) {
    const upr_dims& d = A.d;
    const double* W = A.P->Wee;
    const double* sJ = sh + d.nx + d.nu + 6 * d.nb;
    const double* se = sJ + 6 * d.nq;
    constexpr int nr = ORI ? 6 : 3;   // error rows: position (+ orientation)
    if (!q.terminal) {
        if (lane < NQ) {
            double g = 0.0;
            for (int r = 0; r < nr; ++r) g += W[r] * se[r] * sJ[r * NQ + lane];
            q.out[d.lin_grad + lane] = g;
            for (int m = lane; m < NQ; ++m) {
                double h = 0.0;
                for (int r = 0; r < nr; ++r) h += W[r] * sJ[r * NQ + lane] * sJ[r * NQ + m];
                q.out[d.lin_hess + upr_tri(NQ, lane, m)] = h;
            }
        }
        if (lane == 0) { double c = 0.0; for (int r = 0; r < nr; ++r) c += 0.5 * W[r] * se[r] * se[r]; q.out[d.lin_cost] = c; }
    } else {
        if (lane < 3) q.out[d.lin_grad + lane] = -se[lane];
        if (lane < NQ) for (int r = 0; r < 3; ++r) q.out[d.lin_hess + r * NQ + lane] = sJ[r * NQ + lane];
        if (lane == 0) q.out[d.lin_cost] = 0.0;
    }
}

#ifndef UPR_HOST_EMU
#ifdef UPR_LIN_PROF
// instrumented build (-DUPR_LIN_PROF): cycles per phase, summed over the workgroups by lane 0 (upr_debug_lin_prof reads them)
__device__ unsigned long long upr_lin_prof[8];
#define UPR_LIN_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const long long now_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&upr_lin_prof[i], (unsigned long long)(now_ - t_prof)); t_prof = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define UPR_LIN_STAMP(i) ((void)0)
#endif
// 256 threads = 8 knots x 32 lanes per pass, UPR_LIN_PASSES passes per workgroup.  USE_MFMA: Gauss-Newton Hessian through
// v_mfma_f64_16x16x4_f64.  The one value walk per knot (phase 1a) is a serial chain of ~1.3 k instructions whatever the
// number of active lanes: the workgroup's 8 x PASSES walks run side by side on the first lanes of wave 0, ONCE, and the
// other phases loop over the passes.
// OCC: waves per SIMD the register allocation is held to (the kernel is latency bound: more resident knots hide more of it)
#ifndef UPR_LIN_PASSES
#define UPR_LIN_PASSES 3   // (measured, headline: 1 pass 0.085 ms, 2: 0.079, 3: 0.073, 4: 0.102 -- LDS then allows one workgroup less per CU)
#endif
// NPASS: passes per workgroup (UPR_LIN_PASSES for problems without collision rows; with them the per-knot LDS area is three
// times larger and one pass per workgroup measured best: configs[2] 1.11 ms against 1.70 ms with three passes)
template <int NQ, bool USE_MFMA, int OCC = 2, bool ORI = false, int NPASS = 1>
__global__ void __launch_bounds__(256, OCC) upr_linearize_kernel(upr_lin_args A) {
    extern __shared__ __attribute__((aligned(16))) double smem_all[];
    // the problem record in LDS (upr_linesearch.h: its elements are read all over the phases, the joint frames on the serial
    // chain walk -- out of global memory each is a vector load with its own wait); visible behind the first barrier below
    static_assert(sizeof(upr_problem) % sizeof(double) == 0, "copied as doubles");
    constexpr int NPD = (int)(sizeof(upr_problem) / sizeof(double));
    double* smem = smem_all + ((NPD + 1) & ~1);
    {
        const double* src = reinterpret_cast<const double*>(A.P);
        for (int i = threadIdx.x; i < NPD; i += 256) smem_all[i] = src[i];
    }
    constexpr int NP = UPR_LIN_ANALYTIC ? NPASS : 1;
    const int per = upr_lin_lds_doubles(A.d, A.P->n_sph);
    const int sub = threadIdx.x >> 5, lane = threadIdx.x & 31;
    const int base = blockIdx.x * 8 * NP;
#ifdef UPR_LIN_PROF
    long long t_prof = __builtin_readcyclecounter();
    if (threadIdx.x == 0) atomicAdd(&upr_lin_prof[7], 1ull);
#endif
#if UPR_LIN_SC_ONCE
    {
        static_assert(8 * NP * NQ <= 256, "sincos lanes");
        const int sp = threadIdx.x / NQ, sj = threadIdx.x % NQ;
        if (sp < 8 * NP && base + sp < A.npoints) { const upr_lin_point q = upr_lin_locate(A, base + sp); upr_lin_phase0_sc(A, q, sj, smem + sp * per); }
    }
#endif
#if UPR_LIN_OBS_SNAP && UPR_LIN_ANALYTIC
    if (A.d.no > 0 && A.dyn) {   // the obstacles' states at every knot of the workgroup (read by the rows behind three barriers)
        const int nd = A.P->n_dyn;   // (A.P is still the global record here)
        for (int idx = threadIdx.x; idx < 8 * NP * nd; idx += 256) {
            const int sp = idx / nd, oi = idx - sp * nd;
            if (base + sp < A.npoints) { const upr_lin_point q = upr_lin_locate(A, base + sp); upr_lin_stage_obstacle(A, q, oi, smem + sp * per); }
        }
    }
#endif
#if UPR_LIN_P0_BATCH && UPR_LIN_SC_ONCE && UPR_LIN_ANALYTIC && UPR_LIN_OVERLAP
    // The value walks (a serial chain on 8 NP lanes of wave 0, ~20 k cycles) need sin / cos and the state only: wave 0 walks with
    // the state straight from the input while waves 1 - 3 do the staging and the rows of Df f.
    static_assert(8 * NP <= 64, "walk lanes in wave 0");
    __syncthreads();   // sin / cos, the copy of the problem record
    A.P = reinterpret_cast<const upr_problem*>(smem_all);
    if (threadIdx.x < 64) {
        const int p = base + threadIdx.x;
        if (threadIdx.x < 8 * NP && p < A.npoints) { const upr_lin_point q = upr_lin_locate(A, p); upr_lin_phase1a<NQ>(A, q, 0, smem + threadIdx.x * per, q.x); }
    } else upr_lin_phase0_batched<NQ, NP, 2>(A, base, per, smem);
    __syncthreads();
    UPR_LIN_STAMP(0);
    UPR_LIN_STAMP(1);
#else
#if UPR_LIN_P0_BATCH && UPR_LIN_SC_ONCE
    upr_lin_phase0_batched<NQ, NP>(A, base, per, smem);
#else
#pragma unroll 1
    for (int pp = 0; pp < NP; ++pp) {
        const int slot = pp * 8 + sub, p = base + slot;
        if (p < A.npoints) { const upr_lin_point q = upr_lin_locate(A, p); upr_lin_phase0(A, q, lane, smem + slot * per); }
    }
#endif
    __syncthreads();
    A.P = reinterpret_cast<const upr_problem*>(smem_all);
    UPR_LIN_STAMP(0);
#if UPR_LIN_ANALYTIC
    if (threadIdx.x < 8 * NP) {
        const int p = base + threadIdx.x;
        if (p < A.npoints) { const upr_lin_point q = upr_lin_locate(A, p); upr_lin_phase1a<NQ>(A, q, 0, smem + threadIdx.x * per); }
    }
    __syncthreads();
    UPR_LIN_STAMP(1);
#endif
#endif
#pragma unroll 1
    for (int pp = 0; pp < NP; ++pp) {
        const int slot = pp * 8 + sub, p = base + slot;
        if (p < A.npoints) { const upr_lin_point q = upr_lin_locate(A, p); upr_lin_phase1<NQ, ORI>(A, q, lane, smem + slot * per); }
    }
    UPR_LIN_STAMP(2);
    if (A.d.no > 0) {
#if UPR_LIN_OBS_SNAP && UPR_LIN_ANALYTIC
        // (snapshot form: the 32 lanes of a knot group sit in one wave, so a wave-local ordering point separates placing the
        // spheres from the rows that read them; the group's lanes take (pass, sphere) and (pass, row) jobs of ALL its passes at
        // once -- a problem with five rows per knot keeps ten lanes busy for one trip instead of five lanes for two)
        const int ns = A.P->n_sph, no = A.d.no;
#pragma unroll 1
        for (int idx = lane; idx < NP * ns; idx += UPR_LPK) {
            const int pp = idx / ns, s = idx - pp * ns, slot = pp * 8 + sub, p = base + slot;
            if (p < A.npoints) { const upr_lin_point q = upr_lin_locate(A, p); upr_lin_obs_sphere<NQ>(A, q, s, smem + slot * per); }
        }
        UPR_LIN_STAMP(5);
        UPR_WSYNC();
#pragma unroll 1
        for (int idx = lane; idx < NP * no; idx += UPR_LPK) {
            const int pp = idx / no, r = idx - pp * no, slot = pp * 8 + sub, p = base + slot;
            if (p < A.npoints) { const upr_lin_point q = upr_lin_locate(A, p); upr_lin_obs_row<NQ>(A, q, r, smem + slot * per); }
        }
#else
#pragma unroll 1
        for (int pp = 0; pp < NP; ++pp) {
            const int slot = pp * 8 + sub, p = base + slot;
            if (p < A.npoints) { const upr_lin_point q = upr_lin_locate(A, p); upr_lin_phase_obs_a<NQ>(A, q, lane, smem + slot * per); }
        }
        __syncthreads();
#pragma unroll 1
        for (int pp = 0; pp < NP; ++pp) {
            const int slot = pp * 8 + sub, p = base + slot;
            if (p < A.npoints) { const upr_lin_point q = upr_lin_locate(A, p); upr_lin_phase_obs_b<NQ>(A, q, lane, smem + slot * per); }
        }
#endif
    }
    __syncthreads();
    UPR_LIN_STAMP(3);
#pragma unroll 1
    for (int pp = 0; pp < NP; ++pp) {
        const int slot0 = pp * 8 + sub, p = base + slot0;
        const bool live = p < A.npoints;
        double* sh = smem + slot0 * per;
        upr_lin_point q;
        if (live) q = upr_lin_locate(A, p);
    if (!USE_MFMA) {
        if (live) upr_lin_phase2<NQ, ORI>(A, q, lane, sh);
    } else {
        // One MFMA per knot-half: D(16x16) = A(16x4) B(4x16) with A[i][k] = sqrt(W_k) J[k][i],
        // B[k][j] = sqrt(W_k) J[k][j] (k < 3; k = 3 is zero padding).  Operand lane map (f64 16x16x4,
        // cdna_hip_programming.md section 3): lane L supplies A[L & 15][L >> 4] and B[L >> 4][L & 15];
        // result reg r of lane L is D[(L >> 4) + 4 r][L & 15].  The instruction is wave-wide, so it is
        // issued once per half (h = 0, 1), every lane reading the Jacobian of knot-half h from LDS.
        const int wl = threadIdx.x & 63;       // lane in wave
        const int i16 = wl & 15, k4 = wl >> 4; // operand coordinates
        const int wsub0 = (threadIdx.x >> 6) * 2;  // first knot slot of this wave
        typedef double v4d __attribute__((ext_vector_type(4)));
        for (int h = 0; h < 2; ++h) {
            const int slot = pp * 8 + wsub0 + h;
            const int ph = base + slot;
            const double* shh = smem + slot * per;
            const double* sJ = shh + A.d.nx + A.d.nu + 6 * A.d.nb;
            double a = 0.0;
            if (ph < A.npoints && k4 < 3 && i16 < NQ) a = sqrt(A.P->Wee[k4]) * sJ[k4 * NQ + i16];
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
            if (ORI) {   // the three orientation rows: one more rank-4 update
                double a2 = 0.0;
                if (ph < A.npoints && k4 < 3 && i16 < NQ) a2 = sqrt(A.P->Wee[3 + k4]) * sJ[(3 + k4) * NQ + i16];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, a2, acc, 0, 0, 0);
            }
            if (ph < A.npoints) {
                upr_lin_point qh = upr_lin_locate(A, ph);
                if (!qh.terminal) {
                    const int col = wl & 15;
                    for (int r = 0; r < 4; ++r) {
                        const int row = (wl >> 4) + 4 * r;
                        if (row < NQ && col < NQ && row <= col) qh.out[A.d.lin_hess + upr_tri(NQ, row, col)] = acc[r];
                    }
                }
            }
        }
        // gradient / cost / terminal record stay on the VALU lanes
        if (live) {
            const double* W = A.P->Wee;
            const double* sJ = sh + A.d.nx + A.d.nu + 6 * A.d.nb;
            const double* se = sJ + 6 * A.d.nq;
            constexpr int nr = ORI ? 6 : 3;
            if (!q.terminal) {
                if (lane < NQ) {
                    double g = 0.0;
                    for (int r = 0; r < nr; ++r) g += W[r] * se[r] * sJ[r * NQ + lane];
                    q.out[A.d.lin_grad + lane] = g;
                }
                if (lane == 0) { double c = 0.0; for (int r = 0; r < nr; ++r) c += 0.5 * W[r] * se[r] * se[r]; q.out[A.d.lin_cost] = c; }
            } else {
                upr_lin_phase2<NQ, ORI>(A, q, lane, sh);
            }
        }
    }
    }
    UPR_LIN_STAMP(4);
}
#endif
