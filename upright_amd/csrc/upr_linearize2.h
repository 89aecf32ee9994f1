// upr_linearize2.h -- the linearisation kernel for problems without orientation cost: the same per-knot record as
// upr_linearize.h, produced by lane jobs that are uniform per wave.
//
// What changed against upr_linearize_kernel (which keeps the orientation-cost shapes) and why:
//   * its 896 workgroups of 24 knots held 74 KB of LDS each -- two per CU, 512 resident: a launch of the headline batch ran as
//     TWO rounds (31.7 us at 512 workgroups, 47.3 us at 560, tools/exp_lin_b.py).  Here a knot keeps 213 doubles (no staged
//     x / u, sin / cos and J_p share a slot, the residual's wrench slot holds Df f only, joint 0's snapshot is (o, z) alone: the
//     world in front of it is at rest) and the workgroup copies the 1.6 KB chain PREFIX of the problem record instead of its
//     11 KB: 28 knots = 49 KB, three workgroups per CU, 768 in ONE round;
//   * a tangent lane of the old kernel evaluated the body residual on (value, tangent) pairs with its class -- d/dq, d/dq',
//     d/dq'' -- known only at run time, three classes side by side in every wave.  Here a pass of the workgroup takes ONE class
//     (28 knots x 9 joints = 252 of 256 lanes), with the tangent of the residual written out per class: d/dq'' needs 2 of the
//     12 cross products of d/dq, and no lane multiplies by a zero tangent.
//   * collision / projectile rows: the walk lane places the spheres that ride on a link right behind that link's joint (a list
//     of spheres per frame, built once per workgroup) instead of leaving every link frame in LDS (108 doubles a knot: the knots
//     of a workgroup would drop from 28 to 19); a row is a lane job as before, its gradient out of the snapshots.
// Same closed-form tangents out of the per-joint snapshots (upr_kin.h, "analytic tangents"); results agree with the forward-mode
// walk and the oracle's dual numbers to 1e-10 .. 1e-13 (tests/test_emu.py, tests/test_gpu_parity.py).
#pragma once
#include <cstddef>
#include "upr_linearize.h"

// doubles of the problem record the kernel copies to LDS: dims, dt, the chain (joint frames, tool frame) and gravity
#define UPR_LIN2_NPRE ((int)(offsetof(upr_problem, contact_body1) / sizeof(double)))
static_assert(offsetof(upr_problem, contact_body1) % sizeof(double) == 0, "copied as doubles");

// per-knot LDS area (doubles): gf[ne] = Df f | e[3]: target position, then position error | js: (sin, cos)[nq] during the walk,
// then J_p [3][nq] | the value walk's snapshots | (problems with collision rows) the sphere centres [n_sph][3]
struct upr_lin2_lay { int gf, e, js, snap, obs, per; };
static UPR_HDI upr_lin2_lay upr_lin2_layout(const upr_dims& d, int n_sph) {
    upr_lin2_lay l;
    l.gf = 0; l.e = d.ne; l.js = (l.e + 3 + 1) & ~1; l.snap = l.js + ((3 * d.nq + 1) & ~1);
    l.obs = l.snap + upr_snap_at(d.nq, true) + UPR_SNAP_E;   // (joint 0's snapshot compact: upr_kin.h, COMPACT0)
    l.per = (l.obs + (d.no > 0 ? 3 * n_sph : 0)) | 1;   // (odd: the walk lanes -- one per knot, the same offset in every knot's area -- then hit different LDS banks)
    return l;
}
static UPR_HDI bool upr_lin2_eligible(const upr_lin_args& A) { return A.way_q == nullptr && A.Df != nullptr; }

// Per-workgroup table of the collision spheres (doubles; built once, read by the walk lanes): the spheres that ride on chain
// frames ordered by frame -- start[f] .. start[f + 1] indexes `order` for frame f = 0 .. nq (nq: the tool frame) -- and every
// sphere's offset in its frame.  [start: UPR_MAX_JOINTS + 2 ints | order: UPR_MAX_SPHERES ints | off: UPR_MAX_SPHERES x 3]
#define UPR_LIN2_SPH_START 0
#define UPR_LIN2_SPH_ORDER ((UPR_MAX_JOINTS + 2 + 1) / 2)
#define UPR_LIN2_SPH_OFF (UPR_LIN2_SPH_ORDER + (UPR_MAX_SPHERES + 1) / 2)
#define UPR_LIN2_SPH_DOUBLES (UPR_LIN2_SPH_OFF + 3 * UPR_MAX_SPHERES)
static UPR_HDI int upr_lin2_table_doubles(int n_sph) { return (UPR_LIN2_SPH_OFF + 3 * n_sph + 1) & ~1; }   // (what a workgroup keeps of it)
// dynamic LDS up to which three workgroups share a CU (measured: a launch at 53 760 B runs in one round of 768, at 54 272 B in two
// rounds of 512 -- below the 54 592 B hipOccupancyMaxActiveBlocksPerMultiprocessor allows, tools/probe/lds_occ_probe.hip)
#define UPR_LIN2_LDS_BUDGET 53760
// entry `i` of the table's construction: i < n_sph places sphere i in `order` and copies its offset, i < nq + 2 counts start[i]
static UPR_HDI void upr_lin2_sphere_table(const upr_problem* PG, int nq, int i, double* tab) {
    int* start = reinterpret_cast<int*>(tab + UPR_LIN2_SPH_START); int* order = reinterpret_cast<int*>(tab + UPR_LIN2_SPH_ORDER);
    const int ns = PG->n_sph;
    if (i < ns) {
        const int f = PG->sph_frame[i];
        for (int a = 0; a < 3; ++a) tab[UPR_LIN2_SPH_OFF + 3 * i + a] = PG->sph_off[i][a];
        if (f >= 0) {
            int pos = 0;
            for (int t = 0; t < ns; ++t) { const int ft = PG->sph_frame[t]; if (ft >= 0 && (ft < f || (ft == f && t < i))) ++pos; }
            order[pos] = i;
        }
    }
    if (i < nq + 2) {
        int c = 0;
        for (int t = 0; t < ns; ++t) { const int ft = PG->sph_frame[t]; if (ft >= 0 && ft < i) ++c; }
        start[i] = c;
    }
}
// the walk's hook (upr_kin.h): the spheres of frame f from the frame the walk has just reached
struct upr_lin2_place {
    const double* tab; double* cen;
    UPR_HDI void operator()(int f, const double* C, const double* p) const {
        const int* start = reinterpret_cast<const int*>(tab + UPR_LIN2_SPH_START); const int* order = reinterpret_cast<const int*>(tab + UPR_LIN2_SPH_ORDER);
        for (int t = start[f]; t < start[f + 1]; ++t) {
            const int s = order[t];
            const double* o = tab + UPR_LIN2_SPH_OFF + 3 * s;
            for (int i = 0; i < 3; ++i) cen[3 * s + i] = p[i] + C[3 * i] * o[0] + C[3 * i + 1] * o[1] + C[3 * i + 2] * o[2];
        }
    }
};
// a sphere that does not ride on the chain (world: frame -1; obstacle i: frame -2 - i, at the obstacle's position at the knot)
static UPR_HDI void upr_lin2_job_fixed_sphere(const upr_lin_args& A, const upr_problem* PG, const upr_lin_point& q, int s, double* cen) {
    const int f = PG->sph_frame[s];
    if (f >= 0 || q.terminal) return;
    double ro[3] = {0.0, 0.0, 0.0}, vo[3], ao[3];
    if (f <= -2) upr_lin_obstacle(A, q, -2 - f, ro, vo, ao);
    for (int i = 0; i < 3; ++i) cen[3 * s + i] = ro[i] + PG->sph_off[s][i];
}
// one collision / projectile row of one knot: value, and the gradient in closed form -- a sphere on link f moves rigidly with
// every joint j <= f: d c / d q_j = z_j x (c - o_j) (revolute), z_j (prismatic), 0 (j > f, world and obstacle spheres)
// (P: the record's prefix -- joint types; PG: the whole record -- pairs, radii, projectile data)
template <int NQ>
static UPR_HDI void upr_lin2_job_row(const upr_lin_args& A, const upr_problem* PG, const upr_problem* P, const upr_lin_point& q, int r, const double* snap, const double* cen) {
    const upr_dims& d = A.d;
    if (q.terminal) return;
    // (the projectile rows follow the last obstacle: projectile_path_constraint.h:82 reads state.tail(9))
    double ro[3] = {0.0, 0.0, 0.0}, vo[3] = {0.0, 0.0, 0.0}, ao[3] = {0.0, 0.0, 0.0};
    double flag = 0.0;
    if (A.dyn) {
        upr_lin_obstacle(A, q, PG->n_dyn > 0 ? PG->n_dyn - 1 : 0, ro, vo, ao);
        if (r >= PG->n_pairs) flag = A.pflag ? A.pflag[q.b] : 0.0;
    }
    int sa, sb; double n[3], w;
    q.out[d.lin_obs + r] = upr_state_row(PG, r, [&](int s, int i) { return cen[3 * s + i]; }, ro, vo, ao, flag, &sa, &sb, n, &w);
    const int fa = PG->sph_frame[sa], fb = (sb >= 0) ? PG->sph_frame[sb] : -1;
    double ca[3], cb[3];
    for (int i = 0; i < 3; ++i) { ca[i] = cen[3 * sa + i]; cb[i] = (sb >= 0) ? cen[3 * sb + i] : 0.0; }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const double* S = snap + upr_snap_at(j, true);
        const int zo = (j == 0) ? 3 : 15;
        const double o[3] = {S[0], S[1], S[2]}, z[3] = {S[zo], S[zo + 1], S[zo + 2]};
        double v = 0.0;
        if (P->joint_type[j] == 1) {   // n . (z x (c - o)) = (c - o) . (n x z): the two spheres differ only in c
            double nz[3];
            upr_cross(n, z, nz);
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

// Tangent of the end effector's (C, w, al, a, p) along state coordinate CLS nq + j out of the snapshots (upr_kin.h: the same
// formulas as upr_ee_from_snap, one class at compile time, tangents only).  rot: the direction turns the end effector's frame,
// dC = S(z) C (a revolute joint's angle); dw, dal, da, dp: the other tangents (dp for CLS 0 only).
template <int NQ, int CLS>
static UPR_HDI void upr_lin2_ee_tangent(const upr_problem* P, const double* snap, int j, bool& rot, double* z, double* dw, double* dal, double* da, double* dp) {
    const double* T = snap + upr_snap_at(NQ, true);
    // (joint 0: (o, z) only, the link in front of it is the world -- read as a full snapshot whose other entries are zero)
    double Sv[UPR_SNAP_J];
    {
        const double* Sj = snap + upr_snap_at(j, true);
        if (j == 0) { for (int i = 0; i < UPR_SNAP_J; ++i) Sv[i] = 0.0; for (int i = 0; i < 3; ++i) { Sv[i] = Sj[i]; Sv[15 + i] = Sj[3 + i]; } }
        else for (int i = 0; i < UPR_SNAP_J; ++i) Sv[i] = Sj[i];
    }
    const double* S = Sv;
    const bool rev = P->joint_type[j] == 1;
    double p[3], w[3], o[3], wb[3];
    for (int i = 0; i < 3; ++i) { p[i] = T[9 + i]; w[i] = T[15 + i]; o[i] = S[i]; wb[i] = S[9 + i]; z[i] = S[15 + i]; }
    double rho[3];
    for (int i = 0; i < 3; ++i) rho[i] = p[i] - o[i];
    rot = false;
    for (int i = 0; i < 3; ++i) { dw[i] = 0.0; dal[i] = 0.0; }
    if (CLS == 2) {
        if (rev) { upr_cross(z, rho, da); for (int i = 0; i < 3; ++i) dal[i] = z[i]; }
        else for (int i = 0; i < 3; ++i) da[i] = z[i];
        return;
    }
    double wr[3], t1[3], rd[3];
    upr_cross(wb, rho, t1);
    for (int i = 0; i < 3; ++i) { wr[i] = w[i] - wb[i]; rd[i] = T[12 + i] - S[3 + i] - t1[i]; }   // rho' = v - v_o - w_b x rho
    if (CLS == 1) {
        if (rev) {
            double u[3], wbu[3], wbz[3], zwr[3], zrd[3];
            upr_cross(z, rho, u); upr_cross(wb, u, wbu); upr_cross(wb, z, wbz); upr_cross(z, wr, zwr); upr_cross(z, rd, zrd);
            for (int i = 0; i < 3; ++i) { dw[i] = z[i]; dal[i] = wbz[i] + zwr[i]; da[i] = 2.0 * wbu[i] + 2.0 * zrd[i]; }
        } else {
            double wbz[3];
            upr_cross(wb, z, wbz);
            for (int i = 0; i < 3; ++i) da[i] = 2.0 * wbz[i];
        }
        return;
    }
    // CLS 0
    double ab[3];
    for (int i = 0; i < 3; ++i) ab[i] = S[12 + i];
    if (rev) {
        double t2[3], alr[3], ww[3], t3[3], t4[3], rdd[3];
        upr_cross(wb, wr, t2);
        for (int i = 0; i < 3; ++i) alr[i] = T[21 + i] - ab[i] - t2[i];                       // al_r
        upr_cross(wb, t1, ww);                                                               // w_b x (w_b x rho)
        upr_cross(ab, rho, t3); upr_cross(wb, rd, t4);
        for (int i = 0; i < 3; ++i) rdd[i] = T[18 + i] - S[6 + i] - t3[i] - ww[i] - 2.0 * t4[i];   // rho''
        double u[3], wbu[3], abu[3], wwu[3], zwr[3], wzwr[3], zalr[3], zrd[3], wzrd[3], zrdd[3];
        upr_cross(z, rho, u); upr_cross(wb, u, wbu); upr_cross(ab, u, abu); upr_cross(wb, wbu, wwu);
        upr_cross(z, wr, zwr); upr_cross(wb, zwr, wzwr); upr_cross(z, alr, zalr);
        upr_cross(z, rd, zrd); upr_cross(wb, zrd, wzrd); upr_cross(z, rdd, zrdd);
        for (int i = 0; i < 3; ++i) { dp[i] = u[i]; dw[i] = zwr[i]; dal[i] = wzwr[i] + zalr[i]; da[i] = abu[i] + wwu[i] + 2.0 * wzrd[i] + zrdd[i]; }
        rot = true;
    } else {
        double wbz[3], abz[3], wwz[3];
        upr_cross(wb, z, wbz); upr_cross(ab, z, abz); upr_cross(wb, wbz, wwz);
        for (int i = 0; i < 3; ++i) { dp[i] = z[i]; da[i] = abz[i] + wwz[i]; }
    }
}

// What the residual of one body needs of the end effector's VALUE (contact_constraints.h:80-102; bp: rigid_body.h:36-51):
// r = C c, w x r, acc = a + al x r + w x (w x r) - g, we = C' w, I we, and the residual itself without contact wrench
struct upr_lin2_body { double im, I[9], r[3], wr[3], acc[3], we[3], Iw[3]; };
static UPR_HDI void upr_lin2_body_values(const double* T, const double* bp, const double* g0, upr_lin2_body& V, double* res) {
    const double* C = T; const double* w = T + 15; const double* a = T + 18; const double* al = T + 21;
    const double m = bp[0];
    V.im = upr_rcp(m);   // (hardware reciprocal + one second-order step: no IEEE division sequence per lane and pass)
    const double c[3] = {bp[1] * V.im, bp[2] * V.im, bp[3] * V.im};
    const double I[9] = {bp[4], bp[5], bp[6], bp[5], bp[7], bp[8], bp[6], bp[8], bp[9]};
    for (int i = 0; i < 9; ++i) V.I[i] = I[i];
    upr_rot_const(C, c, V.r);
    double t1[3], t2[3];
    upr_cross(al, V.r, t1); upr_cross(w, V.r, V.wr); upr_cross(w, V.wr, t2);
    for (int i = 0; i < 3; ++i) V.acc[i] = a[i] + t1[i] + t2[i] - g0[i];
    double ae[3];
    for (int i = 0; i < 3; ++i) {
        V.we[i] = C[i] * w[0] + C[3 + i] * w[1] + C[6 + i] * w[2];
        ae[i] = C[i] * al[0] + C[3 + i] * al[1] + C[6 + i] * al[2];
    }
    for (int i = 0; i < 3; ++i) V.Iw[i] = I[3 * i] * V.we[0] + I[3 * i + 1] * V.we[1] + I[3 * i + 2] * V.we[2];
    if (res) {
        double tau[3];
        upr_cross(V.we, V.Iw, tau);
        for (int i = 0; i < 3; ++i) {
            res[i] = V.im * (m * (C[i] * V.acc[0] + C[3 + i] * V.acc[1] + C[6 + i] * V.acc[2]));
            res[3 + i] = V.im * (tau[i] + (I[3 * i] * ae[0] + I[3 * i + 1] * ae[1] + I[3 * i + 2] * ae[2]));
        }
    }
}
// ... and its tangent along a direction (rot, z, dw, dal, da) of class CLS: with dr = z x r (rot),
//   dacc = da + dal x r + al x dr + dw x (w x r) + w x (dw x r + w x dr),  d(C' v) = C' (dv + v x z),
//   d res[0..2] = C' (dacc + acc x z),  d res[3..5] = (d we x I we + we x I d we + I d ae) / m
template <int CLS>
static UPR_HDI void upr_lin2_body_tangent(const double* T, const upr_lin2_body& V, bool rot, const double* z, const double* dw, const double* dal, const double* da, double* out) {
    const double* C = T; const double* w = T + 15; const double* al = T + 21;
    double dacc[3], t[3];
    upr_cross(dal, V.r, t);
    for (int i = 0; i < 3; ++i) dacc[i] = da[i] + t[i];
    double dwe_in[3] = {0.0, 0.0, 0.0}, dae_in[3] = {dal[0], dal[1], dal[2]};
    if (CLS <= 1) {
        double u[3], t2[3];
        upr_cross(dw, V.wr, t); upr_cross(dw, V.r, u); upr_cross(w, u, t2);
        for (int i = 0; i < 3; ++i) { dacc[i] += t[i] + t2[i]; dwe_in[i] = dw[i]; }
    }
    if (CLS == 0 && rot) {
        double dr[3], t2[3], gz[3], wz[3], az[3];
        upr_cross(z, V.r, dr); upr_cross(al, dr, t); upr_cross(w, dr, t2); upr_cross(w, t2, t2);
        upr_cross(V.acc, z, gz); upr_cross(w, z, wz); upr_cross(al, z, az);
        for (int i = 0; i < 3; ++i) { dacc[i] += t[i] + t2[i] + gz[i]; dwe_in[i] += wz[i]; dae_in[i] += az[i]; }
    }
    double dae[3], Idae[3];
    for (int i = 0; i < 3; ++i) {
        out[i] = C[i] * dacc[0] + C[3 + i] * dacc[1] + C[6 + i] * dacc[2];
        dae[i] = C[i] * dae_in[0] + C[3 + i] * dae_in[1] + C[6 + i] * dae_in[2];
    }
    for (int i = 0; i < 3; ++i) Idae[i] = V.I[3 * i] * dae[0] + V.I[3 * i + 1] * dae[1] + V.I[3 * i + 2] * dae[2];
    if (CLS <= 1) {
        double dwe[3], Idwe[3], t1[3], t2[3];
        for (int i = 0; i < 3; ++i) dwe[i] = C[i] * dwe_in[0] + C[3 + i] * dwe_in[1] + C[6 + i] * dwe_in[2];
        for (int i = 0; i < 3; ++i) Idwe[i] = V.I[3 * i] * dwe[0] + V.I[3 * i + 1] * dwe[1] + V.I[3 * i + 2] * dwe[2];
        upr_cross(dwe, V.Iw, t1); upr_cross(V.we, Idwe, t2);
        for (int i = 0; i < 3; ++i) out[3 + i] = V.im * (t1[i] + t2[i] + Idae[i]);
    } else {
        for (int i = 0; i < 3; ++i) out[3 + i] = V.im * Idae[i];
    }
}

// ---- lane jobs (sh: the knot's LDS area; P: the record's PREFIX -- chain and gravity only) ---------------------------------
// one tangent direction of one knot: column CLS nq + j of d g / d x for every body; CLS 0 also column j of J_p
template <int NQ, int CLS>
static UPR_HDI void upr_lin2_job_tangent(const upr_lin_args& A, const upr_problem* P, const upr_lin_point& q, int j, double* sh, const upr_lin2_lay& lay) {
    const upr_dims& d = A.d;
    const double* snap = sh + lay.snap;
    const double* T = snap + upr_snap_at(NQ, true);
    bool rot; double z[3], dw[3], dal[3], da[3], dp[3];
    upr_lin2_ee_tangent<NQ, CLS>(P, snap, j, rot, z, dw, dal, da, dp);
    if (CLS == 0) for (int r = 0; r < 3; ++r) sh[lay.js + r * NQ + j] = dp[r];
    if (q.terminal) return;
    const double* bp = A.body_params + (size_t)q.b * d.nb * 10;
    const int dir = CLS * NQ + j;
    for (int b = 0; b < d.nb; ++b) {
        upr_lin2_body V; double g[6];
        upr_lin2_body_values(T, bp + 10 * b, P->gravity, V, nullptr);
        upr_lin2_body_tangent<CLS>(T, V, rot, z, dw, dal, da, g);
        for (int r = 0; r < 6; ++r) q.out[d.lin_gx + (6 * b + r) * d.nx + dir] = d.eq_scale * g[r];
    }
}
// the residual's value of one body of one knot (the equality is affine in the forces: g = g(x; f = 0) + Df f)
static UPR_HDI void upr_lin2_job_value(const upr_lin_args& A, const upr_problem* P, const upr_lin_point& q, int b, int nq, const double* sh, const upr_lin2_lay& lay) {
    const upr_dims& d = A.d;
    if (q.terminal) return;
    const double* T = sh + lay.snap + upr_snap_at(nq, true);
    upr_lin2_body V; double g[6];
    upr_lin2_body_values(T, A.body_params + ((size_t)q.b * d.nb + b) * 10, P->gravity, V, g);
    for (int r = 0; r < 6; ++r) q.out[d.lin_g + 6 * b + r] = d.eq_scale * g[r] + sh[lay.gf + 6 * b + r];
}
// Df f of one (knot, row) (the forces straight from the input vector)
static UPR_HDI void upr_lin2_job_dff(const upr_lin_args& A, const upr_lin_point& q, int r, double* sh, const upr_lin2_lay& lay) {
    const upr_dims& d = A.d;
    if (q.terminal) return;
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
    sh[lay.gf + r] = v;
}
// gradient entry j (+ the terminal knot's record: grad[0..2] = p_d - p, hess[0..3nq) = J_p); PG: the whole record (weights)
template <int NQ>
static UPR_HDI void upr_lin2_job_grad(const upr_lin_args& A, const upr_problem* PG, const upr_lin_point& q, int j, const double* sh, const upr_lin2_lay& lay) {
    const upr_dims& d = A.d;
    const double* J = sh + lay.js; const double* e = sh + lay.e;
    if (!q.terminal) {
        double g = 0.0;
        for (int r = 0; r < 3; ++r) g += PG->Wee[r] * e[r] * J[r * NQ + j];
        q.out[d.lin_grad + j] = g;
        if (j == 0) { double c = 0.0; for (int r = 0; r < 3; ++r) c += 0.5 * PG->Wee[r] * e[r] * e[r]; q.out[d.lin_cost] = c; }
    } else {
        if (j < 3) q.out[d.lin_grad + j] = -e[j];
        for (int r = 0; r < 3; ++r) q.out[d.lin_hess + r * NQ + j] = J[r * NQ + j];
        if (j == 0) q.out[d.lin_cost] = 0.0;
    }
}
// Gauss-Newton Hessian row j on the VALU (host emulation; the device kernel forms it on the matrix cores)
template <int NQ>
static UPR_HDI void upr_lin2_job_hess_row(const upr_lin_args& A, const upr_problem* PG, const upr_lin_point& q, int j, const double* sh, const upr_lin2_lay& lay) {
    const upr_dims& d = A.d;
    const double* J = sh + lay.js;
    if (q.terminal) return;
    for (int m = j; m < NQ; ++m) {
        double hs = 0.0;
        for (int r = 0; r < 3; ++r) hs += PG->Wee[r] * J[r * NQ + j] * J[r * NQ + m];
        q.out[d.lin_hess + upr_tri(NQ, j, m)] = hs;
    }
}
// everything of one knot, job after job (host emulation; the order of the device kernel's phases)
template <int NQ>
static UPR_HDI void upr_lin2_knot(const upr_lin_args& A, const upr_lin_point& q, double* sh) {
    const upr_dims& d = A.d;
    const upr_lin2_lay lay = upr_lin2_layout(d, A.P->n_sph);
    const upr_problem* P = A.P;
    for (int j = 0; j < NQ; ++j) upr_sincos(q.x[j], sh + lay.js + 2 * j, sh + lay.js + 2 * j + 1);
    if (d.no > 0) {
        double tab[UPR_LIN2_SPH_DOUBLES];
        for (int i = 0; i < UPR_MAX_SPHERES || i < NQ + 2; ++i) upr_lin2_sphere_table(P, NQ, i, tab);
        upr_lin2_place place{tab, sh + lay.obs};
        upr_ee_walk_snap<NQ, upr_lin2_place, true>(P, q.x, sh + lay.js, sh + lay.snap, nullptr, place);
        for (int s2 = 0; s2 < P->n_sph; ++s2) upr_lin2_job_fixed_sphere(A, P, q, s2, sh + lay.obs);
        for (int r = 0; r < d.no; ++r) upr_lin2_job_row<NQ>(A, P, P, q, r, sh + lay.snap, sh + lay.obs);
    } else upr_ee_walk_snap<NQ, upr_no_hook, true>(P, q.x, sh + lay.js, sh + lay.snap, nullptr);
    for (int r = 0; r < d.ne; ++r) upr_lin2_job_dff(A, q, r, sh, lay);
    upr_target_position(P, A.way_p + (size_t)q.b * P->n_way * 3, q.t, sh + lay.e);
    for (int j = 0; j < NQ; ++j) upr_lin2_job_tangent<NQ, 0>(A, P, q, j, sh, lay);
    for (int j = 0; j < NQ; ++j) upr_lin2_job_tangent<NQ, 1>(A, P, q, j, sh, lay);
    for (int j = 0; j < NQ; ++j) upr_lin2_job_tangent<NQ, 2>(A, P, q, j, sh, lay);
    for (int b = 0; b < d.nb; ++b) upr_lin2_job_value(A, P, q, b, NQ, sh, lay);
    const double* T = sh + lay.snap + upr_snap_at(NQ, true);
    for (int r = 0; r < 3; ++r) { sh[lay.e + r] = T[9 + r] - sh[lay.e + r]; if (A.ee_out) A.ee_out[(size_t)q.p * 3 + r] = T[9 + r]; }
    for (int j = 0; j < NQ; ++j) { upr_lin2_job_grad<NQ>(A, P, q, j, sh, lay); upr_lin2_job_hess_row<NQ>(A, P, q, j, sh, lay); }
}

#ifndef UPR_HOST_EMU
#ifdef UPR_LIN_PROF
#define UPR_LIN2_STAMP(i) UPR_LIN_STAMP(i)
#else
#define UPR_LIN2_STAMP(i) ((void)0)
#endif
// 256 lanes, kpw knots per workgroup (the launcher: as many as three workgroups per CU hold in LDS, 28 for the headline shape).
// barrier 1: prefix of the record, sin / cos (a lane per (knot, joint));  wave 0: the value walks, a lane per knot, beside
// waves 1 - 3: Df f (a lane per (knot, row)) and the target positions;  barrier 2;  three tangent passes, a class each, a lane
// per (knot, joint), then the residuals' values and the position errors;  barrier 3;  Hessians on the matrix cores (a knot per
// wave and trip), gradients and costs.
template <int NQ>
__global__ void __launch_bounds__(256, 3) upr_linearize2_kernel(upr_lin_args A, int kpw, int n_sph) {
    extern __shared__ __attribute__((aligned(16))) double smem_all[];
    constexpr int NPRE = UPR_LIN2_NPRE;
    const upr_problem* PG = A.P;
    const upr_dims& d = A.d;
    const upr_lin2_lay lay = upr_lin2_layout(d, n_sph);
    double* tab = smem_all + ((NPRE + 1) & ~1);   // (the sphere table: problems with collision rows only)
    double* smem = tab + (d.no > 0 ? upr_lin2_table_doubles(n_sph) : 0);
    const int tid = threadIdx.x, base = blockIdx.x * kpw;
    const int nk = (A.npoints - base < kpw) ? A.npoints - base : kpw;
#ifdef UPR_LIN_PROF
    long long t_prof = __builtin_readcyclecounter();
    if (threadIdx.x == 0) atomicAdd(&upr_lin_prof[7], 1ull);
#endif
    {
        const double* src = reinterpret_cast<const double*>(PG);
        for (int i = tid; i < NPRE; i += 256) smem_all[i] = src[i];
    }
    static_assert(UPR_MAX_SPHERES >= UPR_MAX_JOINTS + 2, "one lane per entry of the sphere table");
    if (d.no > 0 && tid < UPR_MAX_SPHERES) upr_lin2_sphere_table(PG, NQ, tid, tab);
    for (int job = tid; job < nk * NQ; job += 256) {
        const int s = job / NQ, j = job - s * NQ;
        const upr_lin_point q = upr_lin_locate(A, base + s);
        double s_, c_;
        upr_sincos(q.x[j], &s_, &c_);
        smem[s * lay.per + lay.js + 2 * j] = s_; smem[s * lay.per + lay.js + 2 * j + 1] = c_;
    }
    __syncthreads();
    const upr_problem* P = reinterpret_cast<const upr_problem*>(smem_all);   // (the prefix: chain and gravity only)
    UPR_LIN2_STAMP(0);
    if (tid < 64) {
        if (tid < nk) {
            const upr_lin_point q = upr_lin_locate(A, base + tid);
            double* sh = smem + tid * lay.per;
            if (d.no > 0) { upr_lin2_place place{tab, sh + lay.obs}; upr_ee_walk_snap<NQ, upr_lin2_place, true>(P, q.x, sh + lay.js, sh + lay.snap, nullptr, place); }
            else upr_ee_walk_snap<NQ, upr_no_hook, true>(P, q.x, sh + lay.js, sh + lay.snap, nullptr);
        }
    } else {
        if (d.no > 0) for (int idx = tid - 64; idx < nk * n_sph; idx += 192) {   // the spheres that do not ride on the chain
            const int s = idx / n_sph, sp = idx - s * n_sph;
            const upr_lin_point q = upr_lin_locate(A, base + s);
            upr_lin2_job_fixed_sphere(A, PG, q, sp, smem + s * lay.per + lay.obs);
        }
        for (int idx = tid - 64; idx < nk * d.ne; idx += 192) {
            const int s = idx / d.ne, r = idx - s * d.ne;
            const upr_lin_point q = upr_lin_locate(A, base + s);
            upr_lin2_job_dff(A, q, r, smem + s * lay.per, lay);
        }
        for (int s = tid - 64; s < nk; s += 192) {
            const upr_lin_point q = upr_lin_locate(A, base + s);
            upr_target_position(PG, A.way_p + (size_t)q.b * PG->n_way * 3, q.t, smem + s * lay.per + lay.e);
        }
    }
    __syncthreads();
    UPR_LIN2_STAMP(1);
    for (int job = tid; job < nk * NQ; job += 256) {
        const int s = job / NQ, j = job - s * NQ;
        const upr_lin_point q = upr_lin_locate(A, base + s);
        upr_lin2_job_tangent<NQ, 0>(A, P, q, j, smem + s * lay.per, lay);
    }
    for (int job = tid; job < nk * NQ; job += 256) {
        const int s = job / NQ, j = job - s * NQ;
        const upr_lin_point q = upr_lin_locate(A, base + s);
        upr_lin2_job_tangent<NQ, 1>(A, P, q, j, smem + s * lay.per, lay);
    }
    for (int job = tid; job < nk * NQ; job += 256) {
        const int s = job / NQ, j = job - s * NQ;
        const upr_lin_point q = upr_lin_locate(A, base + s);
        upr_lin2_job_tangent<NQ, 2>(A, P, q, j, smem + s * lay.per, lay);
    }
    UPR_LIN2_STAMP(2);
    for (int job = tid; job < nk * d.nb; job += 256) {
        const int s = job / d.nb, b = job - s * d.nb;
        const upr_lin_point q = upr_lin_locate(A, base + s);
        upr_lin2_job_value(A, P, q, b, NQ, smem + s * lay.per, lay);
    }
    if (d.no > 0) for (int job = tid; job < nk * d.no; job += 256) {   // collision / projectile rows
        const int s = job / d.no, r = job - s * d.no;
        const upr_lin_point q = upr_lin_locate(A, base + s);
        upr_lin2_job_row<NQ>(A, PG, P, q, r, smem + s * lay.per + lay.snap, smem + s * lay.per + lay.obs);
    }
    for (int s = 255 - tid; s < nk; s += 256) {   // (the last lanes: the first ones carry the values)
        double* sh = smem + s * lay.per;
        const double* T = sh + lay.snap + upr_snap_at(NQ, true);
        for (int r = 0; r < 3; ++r) { sh[lay.e + r] = T[9 + r] - sh[lay.e + r]; if (A.ee_out) A.ee_out[(size_t)(base + s) * 3 + r] = T[9 + r]; }
    }
    __syncthreads();
    UPR_LIN2_STAMP(3);
    {
        // One MFMA per knot: D(16x16) = A(16x4) B(4x16) with A[i][k] = sqrt(W_k) J[k][i], B[k][j] = sqrt(W_k) J[k][j] (k < 3;
        // k = 3 is zero padding).  Operand lane map (f64 16x16x4, cdna_hip_programming.md section 3): lane L supplies
        // A[L & 15][L >> 4] and B[L >> 4][L & 15]; result register r of lane L is D[(L >> 4) + 4 r][L & 15].
        const int wl = tid & 63, i16 = wl & 15, k4 = wl >> 4;
        typedef double v4d __attribute__((ext_vector_type(4)));
        const double sw = (k4 < 3) ? sqrt(PG->Wee[k4 < 3 ? k4 : 0]) : 0.0;
        for (int s = tid >> 6; s < nk; s += 4) {
            const double* J = smem + s * lay.per + lay.js;
            const double a = (k4 < 3 && i16 < NQ) ? sw * J[k4 * NQ + i16] : 0.0;
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
            const upr_lin_point q = upr_lin_locate(A, base + s);
            if (!q.terminal) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = k4 + 4 * r;
                    if (row < NQ && i16 < NQ && row <= i16) q.out[d.lin_hess + upr_tri(NQ, row, i16)] = acc[r];
                }
            }
        }
    }
    for (int job = tid; job < nk * NQ; job += 256) {
        const int s = job / NQ, j = job - s * NQ;
        const upr_lin_point q = upr_lin_locate(A, base + s);
        upr_lin2_job_grad<NQ>(A, PG, q, j, smem + s * lay.per, lay);
    }
    UPR_LIN2_STAMP(4);
}
#endif
