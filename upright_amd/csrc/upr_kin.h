// upr_kin.h -- end-effector kinematics of a serial chain and the balancing-constraint math, written
// for one GPU lane per forward-mode tangent direction.
//
// What it replaces: the CppAD tapes the reference evaluates once per knot per SQP iteration --
//   get_rigid_body_state           upright_control/src/constraint/balancing_constraints.cpp:15-30
//   ObjectDynamicsConstraints      upright_control/src/constraint/balancing_constraints.cpp:114-155
//   compute_object_dynamics_*      upright_core/include/upright_core/contact_constraints.h:80-194
//   dC_dtt / skew3                 upright_core/include/upright_core/util.h:27-50
// In the kernel every lane evaluates the same straight-line recursion on a (value, tangent) pair whose
// tangent is seeded with the lane's own state coordinate, so a wave produces a whole Jacobian column
// set in one pass and the 18 x nx kinematic Jacobian never exists in memory.
#pragma once
#include "upr_common.h"

struct upr_dd {  // value + one tangent
    double v, d;
};
static UPR_HDI upr_dd operator+(upr_dd a, upr_dd b) { return {a.v + b.v, a.d + b.d}; }
static UPR_HDI upr_dd operator-(upr_dd a, upr_dd b) { return {a.v - b.v, a.d - b.d}; }
static UPR_HDI upr_dd operator*(upr_dd a, upr_dd b) { return {a.v * b.v, fma(a.d, b.v, a.v * b.d)}; }
static UPR_HDI upr_dd operator*(double a, upr_dd b) { return {a * b.v, a * b.d}; }
static UPR_HDI upr_dd operator*(upr_dd b, double a) { return {a * b.v, a * b.d}; }
static UPR_HDI upr_dd upr_lift(double a, upr_dd*) { return {a, 0.0}; }
static UPR_HDI double upr_lift(double a, double*) { return a; }
static UPR_HDI double upr_seed(double v, bool on, double*) { return v; }
static UPR_HDI upr_dd upr_seed(double v, bool on, upr_dd*) { return {v, on ? 1.0 : 0.0}; }
#ifdef UPR_HOST_EMU
static UPR_HDI void upr_sincos(double th, double* s, double* c) { *s = sin(th); *c = cos(th); }
#else
// (one argument reduction for both: two calls are ~1.6 times the instructions)
static UPR_HDI void upr_sincos(double th, double* s, double* c) { sincos(th, s, c); }
#endif
static UPR_HDI void upr_sincos(upr_dd th, upr_dd* s, upr_dd* c) {
    double sv = sin(th.v), cv = cos(th.v);
    *s = {sv, cv * th.d};
    *c = {cv, -sv * th.d};
}

// the same with sin / cos of the VALUE supplied (computed once per knot instead of by every tangent lane)
static UPR_HDI void upr_sincos_given(double, double sv, double cv, double* s, double* c) { *s = sv; *c = cv; }
static UPR_HDI void upr_sincos_given(upr_dd th, double sv, double cv, upr_dd* s, upr_dd* c) { *s = {sv, cv * th.d}; *c = {cv, -sv * th.d}; }

static UPR_HDI double upr_sqrt_(double a) { return sqrt(a); }
static UPR_HDI upr_dd upr_sqrt_(upr_dd a) { const double r = sqrt(a.v); return {r, 0.5 * a.d / r}; }
static UPR_HDI double upr_recip_(double a) { return 1.0 / a; }
static UPR_HDI upr_dd upr_recip_(upr_dd a) { const double r = 1.0 / a.v; return {r, -a.d * r * r}; }

// Orientation error of the end-effector cost (cost/end_effector_cost.h:42-44,62-66; ocs2 quaternionDistance(q, q_ref) =
// q_w r_v - r_w q_v + q_v x r_v): the vector part of the relative rotation r (x) q^-1, i.e. sin(theta / 2) axis of
// R_d = R_ref C'.  Written on the rotation matrices, which the tangent lanes carry anyway:
// e = vee(R_d - R_d') / (2 sqrt(1 + tr R_d)).  (The sign of a quaternion representative flips e, which cost, gradient
// and Gauss-Newton Hessian do not see.)  C: world <- EE, row-major; Rr: desired, row-major.
static UPR_HDI double upr_val(double a) { return a; }
static UPR_HDI double upr_val(upr_dd a) { return a.v; }
static UPR_HDI double upr_neg(double a) { return -a; }
static UPR_HDI upr_dd upr_neg(upr_dd a) { return {-a.v, -a.d}; }
// vector part of the unit quaternion of Rd, largest-diagonal branch (I, J, K a cyclic permutation): finite where the
// scalar-part branch divides by zero (a relative rotation of 180 degrees); representative with non-negative scalar part
template <int I, int J, int K, class T> static UPR_HDI void upr_quat_vec_diag(const T* Rd, T* e) {
    const T s2 = (Rd[4 * I] - Rd[4 * J]) - Rd[4 * K] + upr_lift(1.0, (T*)nullptr);
    const T s = upr_sqrt_(s2);
    const T is = 0.5 * upr_recip_(s);
    T vi = 0.5 * s, vj = (Rd[3 * J + I] + Rd[3 * I + J]) * is, vk = (Rd[3 * K + I] + Rd[3 * I + K]) * is;
    const T w = (Rd[3 * K + J] - Rd[3 * J + K]) * is;
    if (upr_val(w) < 0.0) { vi = upr_neg(vi); vj = upr_neg(vj); vk = upr_neg(vk); }
    e[I] = vi; e[J] = vj; e[K] = vk;
}
template <class T> static UPR_HDI void upr_orientation_error(const T* C, const double* Rr, T* e) {
    T Rd[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rd[3 * i + j] = Rr[3 * i] * C[3 * j] + Rr[3 * i + 1] * C[3 * j + 1] + Rr[3 * i + 2] * C[3 * j + 2];
    T tr1 = Rd[0] + Rd[4] + Rd[8];
    if (upr_val(tr1) > 0.0) {
        tr1 = tr1 + upr_lift(1.0, (T*)nullptr);
        const T inv = 0.5 * upr_recip_(upr_sqrt_(tr1));
        e[0] = (Rd[7] - Rd[5]) * inv; e[1] = (Rd[2] - Rd[6]) * inv; e[2] = (Rd[3] - Rd[1]) * inv;
    } else {
        // relative rotation beyond 120 degrees (never in a tracking task; a target given upside down gets here): the
        // quaternion by its largest diagonal entry, as ocs2 / Eigen extract it -- the value branch is uniform over the
        // tangent lanes of a knot
        const double d0 = upr_val(Rd[0]), d1 = upr_val(Rd[4]), d2 = upr_val(Rd[8]);
        if (d0 >= d1 && d0 >= d2) upr_quat_vec_diag<0, 1, 2>(Rd, e);
        else if (d1 >= d2) upr_quat_vec_diag<1, 2, 0>(Rd, e);
        else upr_quat_vec_diag<2, 0, 1>(Rd, e);
    }
}

template <class T> static UPR_HDI void upr_cross(const T* a, const T* b, T* r) {
    T r0 = a[1] * b[2] - a[2] * b[1];
    T r1 = a[2] * b[0] - a[0] * b[2];
    T r2 = a[0] * b[1] - a[1] * b[0];
    r[0] = r0; r[1] = r1; r[2] = r2;
}
// r = R * c   (R row-major T, c constant)
template <class T> static UPR_HDI void upr_rot_const(const T* R, const double* c, T* r) {
    for (int i = 0; i < 3; ++i) r[i] = R[3 * i] * c[0] + R[3 * i + 1] * c[1] + R[3 * i + 2] * c[2];
}
// R <- R * M  (M constant 3x3 row-major)
template <class T> static UPR_HDI void upr_rmul_const(T* R, const double* M) {
    for (int i = 0; i < 3; ++i) {
        T a = R[3 * i], b = R[3 * i + 1], c = R[3 * i + 2];
        for (int j = 0; j < 3; ++j) R[3 * i + j] = a * M[j] + b * M[3 + j] + c * M[6 + j];
    }
}
template <class T> static UPR_HDI void upr_rmul(T* R, const T* M) {
    for (int i = 0; i < 3; ++i) {
        T a = R[3 * i], b = R[3 * i + 1], c = R[3 * i + 2];
        for (int j = 0; j < 3; ++j) R[3 * i + j] = a * M[j] + b * M[3 + j] + c * M[6 + j];
    }
}

template <class T>
struct upr_ee {
    T p[3], C[9], v[3], w[3], a[3], al[3];
};

// Rigidly carry the tracked point by the world-frame offset r (classical acceleration).
template <class T> static UPR_HDI void upr_carry(upr_ee<T>& E, const T* r) {
    T wr[3], t[3];
    upr_cross(E.w, r, wr);
    for (int i = 0; i < 3; ++i) E.v[i] = E.v[i] + wr[i];
    upr_cross(E.al, r, t);
    for (int i = 0; i < 3; ++i) E.a[i] = E.a[i] + t[i];
    upr_cross(E.w, wr, t);
    for (int i = 0; i < 3; ++i) { E.a[i] = E.a[i] + t[i]; E.p[i] = E.p[i] + r[i]; }
}

// x: the knot's state [q, v, a] (plain values); dir: tangent direction of this lane (-1: none).
// NQ is the compile-time joint count so that the chain loop unrolls and everything stays in registers.
// sc: optional [NQ][2] = (sin q_j, cos q_j) of the revolute joints, precomputed (NULL: computed here)
// ROLL: the joint loop stays a loop (one joint's code instead of NQ copies with every joint's constants hoisted in front: the
// line search's walk then fits the 128 registers that four of its workgroups per CU leave a lane)
template <class T, int NQ, bool ROLL = false>
static UPR_HDI void upr_ee_kinematics(const upr_problem* P, const double* x, int dir, upr_ee<T>& E, const double* sc = nullptr) {
    T* tag = nullptr;
    for (int i = 0; i < 9; ++i) E.C[i] = upr_lift((i % 4 == 0) ? 1.0 : 0.0, tag);
    for (int i = 0; i < 3; ++i) { E.p[i] = upr_lift(0.0, tag); E.v[i] = E.p[i]; E.w[i] = E.p[i]; E.a[i] = E.p[i]; E.al[i] = E.p[i]; }
#pragma unroll (ROLL ? 1 : NQ)
    for (int j = 0; j < NQ; ++j) {
        T q = upr_seed(x[j], dir == j, tag);
        T qd = upr_seed(x[NQ + j], dir == NQ + j, tag);
        T qdd = upr_seed(x[2 * NQ + j], dir == 2 * NQ + j, tag);
        T r[3];
        upr_rot_const(E.C, P->joint_p[j], r);
        upr_carry(E, r);
        upr_rmul_const(E.C, P->joint_R[j]);
        T z[3];
        upr_rot_const(E.C, P->joint_axis[j], z);
        if (P->joint_type[j] == 1) {
            T wz[3];
            upr_cross(E.w, z, wz);
            for (int i = 0; i < 3; ++i) {
                E.al[i] = E.al[i] + z[i] * qdd + wz[i] * qd;
                E.w[i] = E.w[i] + z[i] * qd;
            }
            // Rodrigues about the constant local axis
            const double* ax = P->joint_axis[j];
            T s, c;
            if (sc) upr_sincos_given(q, sc[2 * j], sc[2 * j + 1], &s, &c); else upr_sincos(q, &s, &c);
            T omc = upr_lift(1.0, tag) - c;
            T M[9];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) M[3 * a + b] = (ax[a] * ax[b]) * omc + ((a == b) ? c : upr_lift(0.0, tag));
            M[1] = M[1] - ax[2] * s; M[2] = M[2] + ax[1] * s;
            M[3] = M[3] + ax[2] * s; M[5] = M[5] - ax[0] * s;
            M[6] = M[6] - ax[1] * s; M[7] = M[7] + ax[0] * s;
            upr_rmul(E.C, M);
        } else {
            T d[3], wz[3], t[3], wd[3];
            for (int i = 0; i < 3; ++i) d[i] = z[i] * q;
            upr_cross(E.w, z, wz);
            upr_cross(E.w, d, wd);
            upr_cross(E.al, d, t);
            for (int i = 0; i < 3; ++i) E.a[i] = E.a[i] + t[i];
            upr_cross(E.w, wd, t);
            for (int i = 0; i < 3; ++i) {
                E.a[i] = E.a[i] + t[i] + (2.0 * wz[i]) * qd + z[i] * qdd;
                E.v[i] = E.v[i] + wd[i] + z[i] * qd;
                E.p[i] = E.p[i] + d[i];
            }
        }
    }
    T r[3];
    upr_rot_const(E.C, P->tool_p, r);
    upr_carry(E, r);
    upr_rmul_const(E.C, P->tool_R);
}

// ---- analytic tangents (round 3) --------------------------------------------------------------------------------------------
// The forward-mode form above walks the chain once per tangent direction on (value, tangent) pairs: 27 walks per knot, every
// one recomputing the same VALUES.  The functions below walk the chain ONCE per knot on plain values, leave a snapshot per
// joint, and every direction's tangent of the end-effector state follows in closed form from the snapshot of its joint:
// everything behind joint j moves RIGIDLY with q_j.  With the base of that motion (the link in front of the joint: origin o,
// its velocity v_o and acceleration a_o as a point of that link, angular velocity w_b, angular acceleration al_b, axis z) and
// the motion of the end effector relative to it
//     rho = p - o,   w_r = w - w_b,   rho' = v - v_o - w_b x rho,   al_r = al - al_b - w_b x w_r,
//     rho'' = a - a_o - al_b x rho - w_b x (w_b x rho) - 2 w_b x rho'
// (the end effector's a = a_o + al_b x rho + w_b x (w_b x rho) + 2 w_b x rho' + rho''), a revolute joint gives
//     d/dq:    dC = S(z) C, dp = z x rho, dw = z x w_r, dal = w_b x (z x w_r) + z x al_r,
//              da = al_b x (z x rho) + w_b x (w_b x (z x rho)) + 2 w_b x (z x rho') + z x rho''
//     d/dq':   dw = z, dal = w_b x z + z x w_r, da = 2 w_b x (z x rho) + 2 z x rho'
//     d/dq'':  dal = z, da = z x rho
// and a prismatic one (everything behind it translates along z)
//     d/dq:    dp = z, da = al_b x z + w_b x (w_b x z);     d/dq':  da = 2 w_b x z;     d/dq'':  da = z.
// Checked against the forward-mode walk and the oracle's dual numbers by the linearisation tests (1e-10 .. 1e-13).
#define UPR_SNAP_J 18   // per joint: o, v_o, a_o, w_b, al_b, z
#define UPR_SNAP_E 24   // end effector: C (9), p, v, w, a, al
// frames (optional, problems with collision spheres): the link frame BEHIND joint j after its motion, [j][C 9, p 3] -- the
// spheres that ride on link j are placed from it (UPR_SNAP_F doubles per joint)
#define UPR_SNAP_F 12
struct upr_no_hook { UPR_HDI void operator()(int, const double*, const double*) const {} };
// hook(f, C, p): called with the link frame behind joint f after its motion (f = NQ: the tool frame) -- upr_linearize2.h places
// the collision spheres that ride on that link there, instead of keeping every frame
// COMPACT0: joint 0's snapshot keeps (o, z) only -- the link in front of the first joint is the world, at rest: v_o = a_o = w_b =
// al_b = 0 exactly -- and the later ones follow it directly (upr_snap_at / UPR_SNAP_C0: upr_linearize2.h, 12 doubles a knot less)
#define UPR_SNAP_C0 6
static UPR_HDI int upr_snap_at(int j, bool compact0) { return compact0 ? (j == 0 ? 0 : UPR_SNAP_C0 + (j - 1) * UPR_SNAP_J) : j * UPR_SNAP_J; }
template <int NQ, class HOOK = upr_no_hook, bool COMPACT0 = false>
static UPR_HDI void upr_ee_walk_snap(const upr_problem* P, const double* x, const double* sc, double* snap, double* frames = nullptr, HOOK hook = HOOK()) {
    upr_ee<double> E;
    for (int i = 0; i < 9; ++i) E.C[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 3; ++i) { E.p[i] = 0.0; E.v[i] = 0.0; E.w[i] = 0.0; E.a[i] = 0.0; E.al[i] = 0.0; }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const double q = x[j], qd = x[NQ + j], qdd = x[2 * NQ + j];
        double r[3];
        upr_rot_const(E.C, P->joint_p[j], r);
        upr_carry(E, r);
        upr_rmul_const(E.C, P->joint_R[j]);
        double z[3];
        upr_rot_const(E.C, P->joint_axis[j], z);
        double* S = snap + upr_snap_at(j, COMPACT0);
        if (COMPACT0 && j == 0) { for (int i = 0; i < 3; ++i) { S[i] = E.p[i]; S[3 + i] = z[i]; } }
        else for (int i = 0; i < 3; ++i) { S[i] = E.p[i]; S[3 + i] = E.v[i]; S[6 + i] = E.a[i]; S[9 + i] = E.w[i]; S[12 + i] = E.al[i]; S[15 + i] = z[i]; }
        if (P->joint_type[j] == 1) {
            double wz[3];
            upr_cross(E.w, z, wz);
            for (int i = 0; i < 3; ++i) { E.al[i] = E.al[i] + z[i] * qdd + wz[i] * qd; E.w[i] = E.w[i] + z[i] * qd; }
            const double* ax = P->joint_axis[j];
            double s_, c_;
            if (sc) { s_ = sc[2 * j]; c_ = sc[2 * j + 1]; } else upr_sincos(q, &s_, &c_);
            const double omc = 1.0 - c_;
            double M[9];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) M[3 * a + b] = (ax[a] * ax[b]) * omc + ((a == b) ? c_ : 0.0);
            M[1] = M[1] - ax[2] * s_; M[2] = M[2] + ax[1] * s_;
            M[3] = M[3] + ax[2] * s_; M[5] = M[5] - ax[0] * s_;
            M[6] = M[6] - ax[1] * s_; M[7] = M[7] + ax[0] * s_;
            upr_rmul(E.C, M);
        } else {
            double d[3], wz[3], t[3], wd[3];
            for (int i = 0; i < 3; ++i) d[i] = z[i] * q;
            upr_cross(E.w, z, wz);
            upr_cross(E.w, d, wd);
            upr_cross(E.al, d, t);
            for (int i = 0; i < 3; ++i) E.a[i] = E.a[i] + t[i];
            upr_cross(E.w, wd, t);
            for (int i = 0; i < 3; ++i) {
                E.a[i] = E.a[i] + t[i] + (2.0 * wz[i]) * qd + z[i] * qdd;
                E.v[i] = E.v[i] + wd[i] + z[i] * qd;
                E.p[i] = E.p[i] + d[i];
            }
        }
        if (frames) {
            double* Fj = frames + j * UPR_SNAP_F;
            for (int i = 0; i < 9; ++i) Fj[i] = E.C[i];
            for (int i = 0; i < 3; ++i) Fj[9 + i] = E.p[i];
        }
        hook(j, E.C, E.p);
    }
    double r[3];
    upr_rot_const(E.C, P->tool_p, r);
    upr_carry(E, r);
    upr_rmul_const(E.C, P->tool_R);
    hook(NQ, E.C, E.p);
    double* T = snap + upr_snap_at(NQ, COMPACT0);
    for (int i = 0; i < 9; ++i) T[i] = E.C[i];
    for (int i = 0; i < 3; ++i) { T[9 + i] = E.p[i]; T[12 + i] = E.v[i]; T[15 + i] = E.w[i]; T[18 + i] = E.a[i]; T[21 + i] = E.al[i]; }
}
// end-effector state with the tangent along state coordinate dir (dir < 0: values only) out of the snapshots
template <int NQ>
static UPR_HDI void upr_ee_from_snap(const upr_problem* P, const double* snap, int dir, upr_ee<upr_dd>& E) {
    const double* T = snap + NQ * UPR_SNAP_J;
    const int cls = (dir >= 0) ? dir / NQ : 3, j = (dir >= 0) ? dir % NQ : 0;
    const double* S = snap + j * UPR_SNAP_J;
    const bool rev = P->joint_type[j] == 1;
    double C[9], p[3], v[3], w[3], a[3], al[3], o[3], vo[3], ao[3], wb[3], ab[3], z[3];
    for (int i = 0; i < 9; ++i) C[i] = T[i];
    for (int i = 0; i < 3; ++i) { p[i] = T[9 + i]; v[i] = T[12 + i]; w[i] = T[15 + i]; a[i] = T[18 + i]; al[i] = T[21 + i];
                                  o[i] = S[i]; vo[i] = S[3 + i]; ao[i] = S[6 + i]; wb[i] = S[9 + i]; ab[i] = S[12 + i]; z[i] = S[15 + i]; }
    double rho[3], wr[3], t1[3], t2[3], rd[3], alr[3], rdd[3];
    for (int i = 0; i < 3; ++i) { rho[i] = p[i] - o[i]; wr[i] = w[i] - wb[i]; }
    upr_cross(wb, rho, t1);
    for (int i = 0; i < 3; ++i) rd[i] = v[i] - vo[i] - t1[i];                       // rho'
    upr_cross(wb, wr, t2);
    for (int i = 0; i < 3; ++i) alr[i] = al[i] - ab[i] - t2[i];                      // al_r
    upr_cross(wb, t1, t2);                                                           // w_b x (w_b x rho)
    double t3[3], t4[3];
    upr_cross(ab, rho, t3);
    upr_cross(wb, rd, t4);
    for (int i = 0; i < 3; ++i) rdd[i] = a[i] - ao[i] - t3[i] - t2[i] - 2.0 * t4[i];  // rho''
    // u = d p / d q_j
    double u[3], zr[3];
    upr_cross(z, rho, zr);
    for (int i = 0; i < 3; ++i) u[i] = rev ? zr[i] : z[i];
    double dC[9], dp[3], dv[3], dw[3], dal[3], da[3];
    for (int i = 0; i < 9; ++i) dC[i] = 0.0;
    for (int i = 0; i < 3; ++i) { dp[i] = 0.0; dv[i] = 0.0; dw[i] = 0.0; dal[i] = 0.0; da[i] = 0.0; }
    double wbu[3];
    upr_cross(wb, u, wbu);
    if (cls == 0) {
        double abu[3], wwu[3];
        upr_cross(ab, u, abu);
        upr_cross(wb, wbu, wwu);
        // v = v_o + w_b x rho + rho': d/dq = w_b x u (+ z x rho' for a revolute joint: rho' turns with the link)
        for (int i = 0; i < 3; ++i) { dp[i] = u[i]; dv[i] = wbu[i]; da[i] = abu[i] + wwu[i]; }
        if (rev) {
            double zwr[3], wzwr[3], zalr[3], zrd[3], wzrd[3], zrdd[3];
            upr_cross(z, wr, zwr);
            upr_cross(wb, zwr, wzwr);
            upr_cross(z, alr, zalr);
            upr_cross(z, rd, zrd);
            upr_cross(wb, zrd, wzrd);
            upr_cross(z, rdd, zrdd);
            for (int i = 0; i < 3; ++i) { dv[i] = dv[i] + zrd[i]; dw[i] = zwr[i]; dal[i] = wzwr[i] + zalr[i]; da[i] = da[i] + 2.0 * wzrd[i] + zrdd[i]; }
            // dC = S(z) C, column by column (C row-major)
            for (int c = 0; c < 3; ++c) {
                const double c0 = C[c], c1 = C[3 + c], c2 = C[6 + c];
                dC[c] = z[1] * c2 - z[2] * c1; dC[3 + c] = z[2] * c0 - z[0] * c2; dC[6 + c] = z[0] * c1 - z[1] * c0;
            }
        }
    } else if (cls == 1) {
        for (int i = 0; i < 3; ++i) { dv[i] = u[i]; da[i] = 2.0 * wbu[i]; }
        if (rev) {
            double wbz[3], zwr[3], zrd[3];
            upr_cross(wb, z, wbz);
            upr_cross(z, wr, zwr);
            upr_cross(z, rd, zrd);
            for (int i = 0; i < 3; ++i) { dw[i] = z[i]; dal[i] = wbz[i] + zwr[i]; da[i] = da[i] + 2.0 * zrd[i]; }
        }
    } else if (cls == 2) {
        for (int i = 0; i < 3; ++i) { da[i] = u[i]; dal[i] = rev ? z[i] : 0.0; }
    }
    for (int i = 0; i < 9; ++i) E.C[i] = {C[i], dC[i]};
    for (int i = 0; i < 3; ++i) { E.p[i] = {p[i], dp[i]}; E.v[i] = {v[i], dv[i]}; E.w[i] = {w[i], dw[i]}; E.a[i] = {a[i], da[i]}; E.al[i] = {al[i], dal[i]}; }
}

// Centres of the collision spheres (controller_interface.cpp:172-228: the spheres of
// upright_assets/thing/xacro/collision_links.urdf.xacro ride on chain links, obstacle spheres are fixed in the
// world): the same chain walk, positions only.  `put(s, c)` receives sphere s and its centre c[3].
template <class T, int NQ, class F>
static UPR_HDI void upr_sphere_walk(const upr_problem* P, const double* x, int dir, F put) {
    T* tag = nullptr;
    T R[9], o[3];
    for (int i = 0; i < 9; ++i) R[i] = upr_lift((i % 4 == 0) ? 1.0 : 0.0, tag);
    for (int i = 0; i < 3; ++i) o[i] = upr_lift(0.0, tag);
    const int ns = P->n_sph;
    auto place = [&](int frame) {
        for (int s = 0; s < ns; ++s) if (P->sph_frame[s] == frame) {
            T c[3];
            upr_rot_const(R, P->sph_off[s], c);
            for (int i = 0; i < 3; ++i) c[i] = c[i] + o[i];
            put(s, c);
        }
    };
    place(-1);
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        T q = upr_seed(x[j], dir == j, tag);
        T r[3];
        upr_rot_const(R, P->joint_p[j], r);
        for (int i = 0; i < 3; ++i) o[i] = o[i] + r[i];
        upr_rmul_const(R, P->joint_R[j]);
        if (P->joint_type[j] == 1) {
            const double* ax = P->joint_axis[j];
            T s, c;
            upr_sincos(q, &s, &c);
            T omc = upr_lift(1.0, tag) - c;
            T M[9];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) M[3 * a + b] = (ax[a] * ax[b]) * omc + ((a == b) ? c : upr_lift(0.0, tag));
            M[1] = M[1] - ax[2] * s; M[2] = M[2] + ax[1] * s;
            M[3] = M[3] + ax[2] * s; M[5] = M[5] - ax[0] * s;
            M[6] = M[6] - ax[1] * s; M[7] = M[7] + ax[0] * s;
            upr_rmul(R, M);
        } else {
            T z[3];
            upr_rot_const(R, P->joint_axis[j], z);
            for (int i = 0; i < 3; ++i) o[i] = o[i] + z[i] * q;
        }
        place(j);
    }
    T r[3];
    upr_rot_const(R, P->tool_p, r);
    for (int i = 0; i < 3; ++i) o[i] = o[i] + r[i];
    upr_rmul_const(R, P->tool_R);
    place(NQ);
}
// closest FUTURE time of a ballistic path r0 + t v0 + t^2/2 g to the point c (projectile_path_constraint.h:12-45:
// stationary point of the squared distance by Newton on the cubic, 10 steps from t = 0, tolerance 1e-4)
static UPR_HDI double upr_projectile_closest_time(const double* c, const double* r0, const double* v0, const double* g) {
    const double dr[3] = {c[0] - r0[0], c[1] - r0[1], c[2] - r0[2]};
    const double a = g[0] * g[0] + g[1] * g[1] + g[2] * g[2], b = 3.0 * (v0[0] * g[0] + v0[1] * g[1] + v0[2] * g[2]);
    const double cc = 2.0 * (v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2] - (dr[0] * g[0] + dr[1] * g[1] + dr[2] * g[2]));
    const double dd = -2.0 * (dr[0] * v0[0] + dr[1] * v0[1] + dr[2] * v0[2]);
    double x = 0.0;
    for (int i = 0; i < 10; ++i) {
        const double f = ((a * x + b) * x + cc) * x + dd, df = (3.0 * a * x + 2.0 * b) * x + cc;
        const double upd = f * upr_rcp(df);   // (hardware reciprocal + one second-order step: the IEEE sequence is ~100 dependent cycles per trip)
        x -= upd;
        if (fabs(upd) < 1e-4) break;
    }
    return x > 0.0 ? x : 0.0;
}
// obstacle state tau seconds after its observation xo = [r, v, a]
static UPR_HDI void upr_obstacle_at(const double* xo, double tau, double* r, double* v, double* a) {
    for (int i = 0; i < 3; ++i) { a[i] = xo[6 + i]; v[i] = xo[3 + i] + tau * a[i]; r[i] = xo[i] + tau * xo[3 + i] + 0.5 * tau * tau * a[i]; }
}
// one state row from the sphere centres: collision pair / ground row r < n_pairs, projectile row otherwise.
// cen(s, i) returns coordinate i of the centre of sphere s; returns the value and the unit direction n (the row's
// gradient is wgt * n . d c_a/dq - wgt * n . d c_b/dq, b < 0: no second sphere)
template <class CEN>
static UPR_HDI double upr_state_row(const upr_problem* P, int r, CEN cen, const double* ro, const double* vo, const double* ao, double flag,
                                          int* sa, int* sb, double* n, double* wgt) {
    if (r < P->n_pairs) {
        const int a = P->pair_a[r], b = P->pair_b[r];
        *sa = a; *sb = b; *wgt = 1.0;
        if (b < 0) { n[0] = 0.0; n[1] = 0.0; n[2] = 1.0; return cen(a, 2) - (P->sph_r[a] + P->obs_min_dist); }
        double e[3] = {cen(a, 0) - cen(b, 0), cen(a, 1) - cen(b, 1), cen(a, 2) - cen(b, 2)};
        // |e| and e / |e| from one reciprocal square root (relative error < 3e-16) instead of a square root and three divisions
        const double ss = e[0] * e[0] + e[1] * e[1] + e[2] * e[2], ri = upr_rsqrt(ss), dist = ss * ri;
        for (int i = 0; i < 3; ++i) n[i] = e[i] * ri;
        return dist - (P->sph_r[a] + P->sph_r[b] + P->obs_min_dist);
    }
    const int i = r - P->n_pairs, a = P->proj_sph[i];
    *sa = a; *sb = -1;
    const double c[3] = {cen(a, 0), cen(a, 1), cen(a, 2)};
    const double dt = (flag > 0.5) ? upr_projectile_closest_time(c, ro, vo, ao) : 0.0;
    double e[3];
    for (int j = 0; j < 3; ++j) e[j] = c[j] - (ro[j] + dt * vo[j] + 0.5 * dt * dt * ao[j]);
    const double ss = e[0] * e[0] + e[1] * e[1] + e[2] * e[2], ri = upr_rsqrt(ss), dist = ss * ri;
    for (int j = 0; j < 3; ++j) n[j] = e[j] * ri;
    *wgt = flag * P->proj_scale / P->proj_dist[i];
    return *wgt * (dist - P->proj_dist[i]);
}
// values of the state rows at a configuration (line search): d[n_pairs + n_proj]; xo: the obstacles' states at this knot,
// [n_dyn][9] (the projectile rows follow the last one: projectile_path_constraint.h:82 reads state.tail(9))
template <int NQ>
static UPR_HDI void upr_obstacle_values(const upr_problem* P, const double* x, const double* xo, double flag, double* d) {
    double c[UPR_MAX_SPHERES][3];
    double ro[3] = {0, 0, 0}, vo[3] = {0, 0, 0}, ao[3] = {0, 0, 0};
    if (xo && P->n_dyn > 0) upr_obstacle_at(xo + 9 * (P->n_dyn - 1), 0.0, ro, vo, ao);
    for (int s = 0; s < P->n_sph; ++s) if (P->sph_frame[s] <= -2) for (int i = 0; i < 3; ++i) c[s][i] = (xo ? xo[9 * (-2 - P->sph_frame[s]) + i] : 0.0) + P->sph_off[s][i];
    upr_sphere_walk<double, NQ>(P, x, -1, [&](int s, const double* cs) { c[s][0] = cs[0]; c[s][1] = cs[1]; c[s][2] = cs[2]; });
    for (int r = 0; r < P->n_pairs + P->n_proj; ++r) {
        int sa, sb; double n[3], w;
        d[r] = upr_state_row(P, r, [&](int s, int i) { return c[s][i]; }, ro, vo, ao, flag, &sa, &sb, n, &w);
    }
}

// Object-dynamics residual of body b (contact_constraints.h:80-102), unnormalised, given the summed
// contact wrench (F, Tq) on the body (plain values: the wrench does not depend on the state).
// bp = the body's 10 inertial parameters (rigid_body.h:36-51).
template <class T>
static UPR_HDI void upr_body_residual(const upr_ee<T>& E, const double* bp, const double* g0, const double* F,
                                            const double* Tq, T* out) {
    T* tag = nullptr;
    const double m = bp[0], im = 1.0 / m;   // (one division: the centre of mass by the reciprocal)
    const double c[3] = {bp[1] * im, bp[2] * im, bp[3] * im};
    const double I[9] = {bp[4], bp[5], bp[6], bp[5], bp[7], bp[8], bp[6], bp[8], bp[9]};
    // ddC c = (S(al) + S(w) S(w)) (C c) = al x r + w x (w x r), r = C c      util.h:39-44
    T r[3], t1[3], t2[3], acc[3];
    upr_rot_const(E.C, c, r);
    upr_cross(E.al, r, t1);
    upr_cross(E.w, r, t2);
    upr_cross(E.w, t2, t2);
    for (int i = 0; i < 3; ++i) acc[i] = E.a[i] + t1[i] + t2[i] - upr_lift(g0[i], tag);
    // C^T (.)
    T we[3], ae[3], gif[3];
    for (int i = 0; i < 3; ++i) {
        gif[i] = m * (E.C[i] * acc[0] + E.C[3 + i] * acc[1] + E.C[6 + i] * acc[2]);
        we[i] = E.C[i] * E.w[0] + E.C[3 + i] * E.w[1] + E.C[6 + i] * E.w[2];
        ae[i] = E.C[i] * E.al[0] + E.C[3 + i] * E.al[1] + E.C[6 + i] * E.al[2];
    }
    T Iw[3], Ia[3], tau[3];
    for (int i = 0; i < 3; ++i) {
        Iw[i] = I[3 * i] * we[0] + I[3 * i + 1] * we[1] + I[3 * i + 2] * we[2];
        Ia[i] = I[3 * i] * ae[0] + I[3 * i + 1] * ae[1] + I[3 * i + 2] * ae[2];
    }
    upr_cross(we, Iw, tau);
    for (int i = 0; i < 3; ++i) {
        out[i] = im * (gif[i] - upr_lift(F[i], tag));
        out[3 + i] = im * (tau[i] + Ia[i] - upr_lift(Tq[i], tag));
    }
}

// contact_constraints.h:107-157: summed contact wrench on every balanced body.  forces = u tail.
// Fw[nb][6] = [F(3), T(3)].  bodies: body_params[nb][10].
static UPR_HDI void upr_object_wrenches(const upr_problem* P, const double* body_params, const double* forces,
                                              double* Fw) {
    for (int i = 0; i < 6 * P->nb; ++i) Fw[i] = 0.0;
    for (int i = 0; i < P->nc; ++i) {
        double f[3];
        if (P->nf == 1) { for (int a = 0; a < 3; ++a) f[a] = forces[i] * P->contact_normal[i][a]; }
        else { f[0] = forces[3 * i]; f[1] = forces[3 * i + 1]; f[2] = forces[3 * i + 2]; }
        int b1 = P->contact_body1[i], b2 = P->contact_body2[i];
        if (b1 >= 0) {
            const double* bp = body_params + 10 * b1;
            double l[3] = {P->contact_r1[i][0] - bp[1] / bp[0], P->contact_r1[i][1] - bp[2] / bp[0], P->contact_r1[i][2] - bp[3] / bp[0]};
            double* W = Fw + 6 * b1;
            W[0] += f[0]; W[1] += f[1]; W[2] += f[2];
            W[3] += l[1] * f[2] - l[2] * f[1]; W[4] += l[2] * f[0] - l[0] * f[2]; W[5] += l[0] * f[1] - l[1] * f[0];
        }
        {
            const double* bp = body_params + 10 * b2;
            double l[3] = {P->contact_r2[i][0] - bp[1] / bp[0], P->contact_r2[i][1] - bp[2] / bp[0], P->contact_r2[i][2] - bp[3] / bp[0]};
            double* W = Fw + 6 * b2;
            W[0] -= f[0]; W[1] -= f[1]; W[2] -= f[2];
            W[3] -= l[1] * f[2] - l[2] * f[1]; W[4] -= l[2] * f[0] - l[0] * f[2]; W[5] -= l[0] * f[1] - l[1] * f[0];
        }
    }
}

// The same sums for arrangements with at most NCM contact points, every operand requested before the first use (the
// loop above exposes one memory latency per contact and per operand group); same order of additions, bitwise the same result.
template <int NCM>
static UPR_HDI void upr_object_wrenches_small(const upr_problem* P, const double* body_params, const double* forces, double* Fw) {
    const int nc = P->nc, nf = P->nf, nb = P->nb;
    double f[NCM][3], r1[NCM][3], r2[NCM][3], c1[NCM][3], c2[NCM][3];
    int b1[NCM], b2[NCM];
#pragma unroll
    for (int i = 0; i < NCM; ++i) {
        const int ii = (i < nc) ? i : 0;
        b1[i] = P->contact_body1[ii]; b2[i] = P->contact_body2[ii];
        if (nf == 1) { const double s = forces[ii]; for (int a = 0; a < 3; ++a) f[i][a] = s * P->contact_normal[ii][a]; }
        else { f[i][0] = forces[3 * ii]; f[i][1] = forces[3 * ii + 1]; f[i][2] = forces[3 * ii + 2]; }
        for (int a = 0; a < 3; ++a) { r1[i][a] = P->contact_r1[ii][a]; r2[i][a] = P->contact_r2[ii][a]; }
    }
#pragma unroll
    for (int i = 0; i < NCM; ++i) {
        const double* bq = body_params + 10 * b2[i];
        const double* bp = body_params + 10 * (b1[i] >= 0 ? b1[i] : 0);
        for (int a = 0; a < 3; ++a) { c2[i][a] = bq[1 + a] / bq[0]; c1[i][a] = bp[1 + a] / bp[0]; }
    }
    for (int i = 0; i < 6 * nb; ++i) Fw[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NCM; ++i) {
        if (i >= nc) continue;
        if (b1[i] >= 0) {
            const double l[3] = {r1[i][0] - c1[i][0], r1[i][1] - c1[i][1], r1[i][2] - c1[i][2]};
            double* W = Fw + 6 * b1[i];
            W[0] += f[i][0]; W[1] += f[i][1]; W[2] += f[i][2];
            W[3] += l[1] * f[i][2] - l[2] * f[i][1]; W[4] += l[2] * f[i][0] - l[0] * f[i][2]; W[5] += l[0] * f[i][1] - l[1] * f[i][0];
        }
        {
            const double l[3] = {r2[i][0] - c2[i][0], r2[i][1] - c2[i][1], r2[i][2] - c2[i][2]};
            double* W = Fw + 6 * b2[i];
            W[0] -= f[i][0]; W[1] -= f[i][1]; W[2] -= f[i][2];
            W[3] -= l[1] * f[i][2] - l[2] * f[i][1]; W[4] -= l[2] * f[i][0] - l[0] * f[i][2]; W[5] -= l[0] * f[i][1] - l[1] * f[i][0];
        }
    }
}

// contact_constraints.h:50-77: the five pyramid rows of contact i
static UPR_HDI void upr_friction_rows_contact(const upr_problem* P, int i, const double* f, double* h) {
    const double* n = P->contact_normal[i];
    const double* S = P->contact_span[i];
    double fn = n[0] * f[0] + n[1] * f[1] + n[2] * f[2];
    double t0 = S[0] * f[0] + S[1] * f[1] + S[2] * f[2];
    double t1 = S[3] * f[0] + S[4] * f[1] + S[5] * f[2];
    double mu = P->contact_mu[i];
    h[0] = fn;
    h[1] = mu * fn - t0 - t1;
    h[2] = mu * fn - t0 + t1;
    h[3] = mu * fn + t0 - t1;
    h[4] = mu * fn + t0 + t1;
}
// row r (0..4) of the constant 5x3 Jacobian d h / d f of contact i
static UPR_HDI void upr_friction_row_jac(const upr_problem* P, int i, int r, double* e) {
    const double* n = P->contact_normal[i];
    const double* S = P->contact_span[i];
    double mu = P->contact_mu[i];
    if (r == 0) { e[0] = n[0]; e[1] = n[1]; e[2] = n[2]; return; }
    double s0 = (r == 1 || r == 2) ? -1.0 : 1.0;
    double s1 = (r == 1 || r == 3) ? -1.0 : 1.0;
    for (int a = 0; a < 3; ++a) e[a] = mu * n[a] + s0 * S[a] + s1 * S[3 + a];
}
