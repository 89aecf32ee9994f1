// upright_oracle.cpp -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
// See upright_oracle.h for the parity status of every piece.  Plain C++17, fp64, no Eigen.
//
// Reference map (paths relative to /root/reference):
//   rigid-body math      upright_core/include/upright_core/contact_constraints.h:50-194,
//                        rigid_body.h:11-51, util.h:27-50
//   constraint wrappers  upright_control/src/constraint/balancing_constraints.cpp:15-155
//   OCP assembly         upright_control/src/controller_interface.cpp:103-420
//   costs                upright_control/include/upright_control/cost/end_effector_cost.h:33-84,
//                        cost/quadratic_joint_state_input_cost.h:9-34
//   terminal constraint  constraint/stationary_desired_position_constraint.h:35-74
//   dynamics             dynamics/system_dynamics.h:15-22 (triple integrator, integrated exactly)
//   SQP / QP / FK        [UPSTREAM, absent] ocs2_sqp MultipleShootingSolver, HPIPM, Pinocchio:
//                        restated from their published algorithms ("parity unpinned").
#include "upright_oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------------
// forward-mode dual number (one tangent direction at a time)
struct Dual {
    double v, d;
    Dual() : v(0), d(0) {}
    Dual(double v_) : v(v_), d(0) {}
    Dual(double v_, double d_) : v(v_), d(d_) {}
};
inline Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
inline Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
inline Dual operator-(Dual a) { return {-a.v, -a.d}; }
inline Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
inline Dual operator/(Dual a, Dual b) { return {a.v / b.v, (a.d * b.v - a.v * b.d) / (b.v * b.v)}; }
inline Dual sin(Dual a) { return {std::sin(a.v), std::cos(a.v) * a.d}; }
inline Dual cos(Dual a) { return {std::cos(a.v), -std::sin(a.v) * a.d}; }
inline Dual sqrt(Dual a) { const double r = std::sqrt(a.v); return {r, 0.5 * a.d / r}; }
inline double sqrt(double a) { return std::sqrt(a); }
inline double sin(double a) { return std::sin(a); }
inline double cos(double a) { return std::cos(a); }
inline double val(double a) { return a; }
inline double val(Dual a) { return a.v; }

template <class T>
struct V3 {
    T x[3];
    T& operator[](int i) { return x[i]; }
    const T& operator[](int i) const { return x[i]; }
};
template <class T>
struct M3 {
    T m[9];
    T& operator()(int i, int j) { return m[3 * i + j]; }
    const T& operator()(int i, int j) const { return m[3 * i + j]; }
};
template <class T> V3<T> add(const V3<T>& a, const V3<T>& b) { return {{a[0] + b[0], a[1] + b[1], a[2] + b[2]}}; }
template <class T> V3<T> sub(const V3<T>& a, const V3<T>& b) { return {{a[0] - b[0], a[1] - b[1], a[2] - b[2]}}; }
template <class T> V3<T> scl(const V3<T>& a, T s) { return {{a[0] * s, a[1] * s, a[2] * s}}; }
template <class T> T dot(const V3<T>& a, const V3<T>& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <class T> V3<T> cross(const V3<T>& a, const V3<T>& b) {
    return {{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}};
}
template <class T> V3<T> mv(const M3<T>& A, const V3<T>& b) {
    V3<T> r;
    for (int i = 0; i < 3; ++i) r[i] = A(i, 0) * b[0] + A(i, 1) * b[1] + A(i, 2) * b[2];
    return r;
}
template <class T> V3<T> mtv(const M3<T>& A, const V3<T>& b) {  // A^T b
    V3<T> r;
    for (int i = 0; i < 3; ++i) r[i] = A(0, i) * b[0] + A(1, i) * b[1] + A(2, i) * b[2];
    return r;
}
template <class T> M3<T> mm(const M3<T>& A, const M3<T>& B) {
    M3<T> C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C(i, j) = A(i, 0) * B(0, j) + A(i, 1) * B(1, j) + A(i, 2) * B(2, j);
    return C;
}
template <class T> M3<T> madd(const M3<T>& A, const M3<T>& B) {
    M3<T> C;
    for (int i = 0; i < 9; ++i) C.m[i] = A.m[i] + B.m[i];
    return C;
}
// upright_core/include/upright_core/util.h:27-36
template <class T> M3<T> skew3(const V3<T>& x) {
    M3<T> M;
    M(0, 0) = T(0.0); M(0, 1) = -x[2];   M(0, 2) = x[1];
    M(1, 0) = x[2];   M(1, 1) = T(0.0);  M(1, 2) = -x[0];
    M(2, 0) = -x[1];  M(2, 1) = x[0];    M(2, 2) = T(0.0);
    return M;
}
template <class T> M3<T> from_d(const double* r) {
    M3<T> M;
    for (int i = 0; i < 9; ++i) M.m[i] = T(r[i]);
    return M;
}
template <class T> V3<T> from_d3(const double* r) { return {{T(r[0]), T(r[1]), T(r[2])}}; }

// Rodrigues rotation about a constant unit axis
template <class T> M3<T> axis_rot(const double* ax, T th) {
    T c = cos(th), s = sin(th);
    T one_c = T(1.0) - c;
    M3<T> R;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R(i, j) = T(ax[i] * ax[j]) * one_c + (i == j ? c : T(0.0));
    R(0, 1) = R(0, 1) - T(ax[2]) * s; R(0, 2) = R(0, 2) + T(ax[1]) * s;
    R(1, 0) = R(1, 0) + T(ax[2]) * s; R(1, 2) = R(1, 2) - T(ax[0]) * s;
    R(2, 0) = R(2, 0) - T(ax[1]) * s; R(2, 1) = R(2, 1) + T(ax[0]) * s;
    return R;
}

// upright_core/include/upright_core/types.h:31-85 RigidBodyState (world frame, EE origin)
template <class T>
struct EEState {
    V3<T> p; M3<T> C; V3<T> v, w, a, al;
};

// EE kinematics of a serial chain.  Restates what balancing_constraints.cpp:15-30 pulls from
// PinocchioEndEffectorKinematicsCppAd [UPSTREAM]: world-frame position/orientation, velocity,
// angular velocity, CLASSICAL linear acceleration of the frame origin, angular acceleration.
template <class T>
EEState<T> ee_kin(const orc_problem* P, const T* x) {
    const int nq = P->nq;
    M3<T> R; for (int i = 0; i < 9; ++i) R.m[i] = T((i % 4 == 0) ? 1.0 : 0.0);
    V3<T> o{{T(0.0), T(0.0), T(0.0)}}, w = o, al = o, v = o, a = o;
    auto offset = [&](const V3<T>& r) {  // move the tracked point by world offset r (rigidly attached)
        v = add(v, cross(w, r));
        a = add(a, add(cross(al, r), cross(w, cross(w, r))));
        o = add(o, r);
    };
    for (int i = 0; i < nq; ++i) {
        T q = x[i], qd = x[nq + i], qdd = x[2 * nq + i];
        offset(mv(R, from_d3<T>(P->joint_p[i])));
        R = mm(R, from_d<T>(P->joint_R[i]));
        V3<T> z = mv(R, from_d3<T>(P->joint_axis[i]));
        if (P->joint_type[i] == 1) {
            al = add(al, add(scl(z, qdd), scl(cross(w, z), qd)));
            w = add(w, scl(z, qd));
            R = mm(R, axis_rot<T>(P->joint_axis[i], q));
        } else {
            V3<T> d = scl(z, q);
            V3<T> wz = cross(w, z);
            a = add(a, add(add(cross(al, d), cross(w, cross(w, d))), add(scl(wz, T(2.0) * qd), scl(z, qdd))));
            v = add(v, add(cross(w, d), scl(z, qd)));
            o = add(o, d);
        }
    }
    offset(mv(R, from_d3<T>(P->tool_p)));
    R = mm(R, from_d<T>(P->tool_R));
    EEState<T> S;
    S.p = o; S.C = R; S.v = v; S.w = w; S.a = a; S.al = al;
    return S;
}

// centres of the collision spheres: the same chain walk, positions only
template <class T>
void sphere_centers(const orc_problem* P, const T* x, double tau, std::vector<V3<T>>& c) {
    const int nq = P->nq;
    c.assign(P->n_sph, V3<T>{{T(0.0), T(0.0), T(0.0)}});
    for (int s = 0; s < P->n_sph; ++s) if (P->sph_frame[s] <= -2) {   // rides on dynamic obstacle -2 - frame: ballistic, independent of the robot
        const double* xo = P->dyn_x0 + 9 * (-2 - P->sph_frame[s]);
        for (int i = 0; i < 3; ++i) c[s][i] = T(xo[i] + tau * xo[3 + i] + 0.5 * tau * tau * xo[6 + i] + P->sph_off[s][i]);
    }
    M3<T> R; for (int i = 0; i < 9; ++i) R.m[i] = T((i % 4 == 0) ? 1.0 : 0.0);
    V3<T> o{{T(0.0), T(0.0), T(0.0)}};
    auto place = [&](int frame) {
        for (int s = 0; s < P->n_sph; ++s) if (P->sph_frame[s] == frame) c[s] = add(o, mv(R, from_d3<T>(P->sph_off[s])));
    };
    place(-1);
    for (int i = 0; i < nq; ++i) {
        o = add(o, mv(R, from_d3<T>(P->joint_p[i])));
        R = mm(R, from_d<T>(P->joint_R[i]));
        if (P->joint_type[i] == 1) R = mm(R, axis_rot<T>(P->joint_axis[i], x[i]));
        else o = add(o, scl(mv(R, from_d3<T>(P->joint_axis[i])), x[i]));
        place(i);
    }
    o = add(o, mv(R, from_d3<T>(P->tool_p)));
    R = mm(R, from_d<T>(P->tool_R));
    place(nq);
}
// projectile_path_constraint.h:12-45
inline double cubic_newton(double a, double b, double c, double d, double x0, double tol) {
    double x = x0;
    for (int i = 0; i < 10; ++i) {
        const double f = a * x * x * x + b * x * x + c * x + d, df = 3 * a * x * x + 2 * b * x + c;
        const double upd = f / df;
        x -= upd;
        if (std::fabs(upd) < tol) return x;
    }
    return x;
}
template <class T>
void obstacle_rows(const orc_problem* P, const T* x, double tau, T* d) {
    std::vector<V3<T>> c;
    sphere_centers<T>(P, x, tau, c);
    for (int r = 0; r < P->n_pairs; ++r) {
        const int a = P->pair_a[r], b = P->pair_b[r];
        if (b < 0) { d[r] = c[a][2] - T(P->sph_r[a] + P->obs_min_dist); continue; }   // ground half-space
        V3<T> e = sub(c[a], c[b]);
        d[r] = sqrt(dot(e, e)) - T(P->sph_r[a] + P->sph_r[b] + P->obs_min_dist);
    }
    // projectile path rows (the closest time is held fixed in the derivative, projectile_path_constraint.h:118-145)
    double ro[3], vo[3], ao[3];
    const double* xl = P->dyn_x0 + 9 * (P->n_dyn > 0 ? P->n_dyn - 1 : 0);   // projectile_path_constraint.h:82: state.tail(9), the last obstacle
    for (int i = 0; i < 3; ++i) { ao[i] = xl[6 + i]; vo[i] = xl[3 + i] + tau * ao[i]; ro[i] = xl[i] + tau * xl[3 + i] + 0.5 * tau * tau * ao[i]; }
    for (int i = 0; i < P->n_proj; ++i) {
        const V3<T>& cl = c[P->proj_sph[i]];
        double dt = 0.0;
        if (P->proj_s > 0.5) {
            double dr[3] = {val(cl[0]) - ro[0], val(cl[1]) - ro[1], val(cl[2]) - ro[2]};
            const double gg = ao[0] * ao[0] + ao[1] * ao[1] + ao[2] * ao[2], vg = vo[0] * ao[0] + vo[1] * ao[1] + vo[2] * ao[2];
            const double vv = vo[0] * vo[0] + vo[1] * vo[1] + vo[2] * vo[2], dg = dr[0] * ao[0] + dr[1] * ao[1] + dr[2] * ao[2];
            const double dv = dr[0] * vo[0] + dr[1] * vo[1] + dr[2] * vo[2];
            dt = std::max(0.0, cubic_newton(gg, 3 * vg, 2 * (vv - dg), -2 * dv, 0.0, 1e-4));
        }
        V3<T> e;
        for (int j = 0; j < 3; ++j) e[j] = cl[j] - T(ro[j] + dt * vo[j] + 0.5 * dt * dt * ao[j]);
        const double w = P->proj_scale / P->proj_dist[i];
        d[P->n_pairs + i] = T(w * P->proj_s) * (sqrt(dot(e, e)) - T(P->proj_dist[i]));
    }
}

// rigid_body.h:36-43 from_parameters
template <class T>
struct Body { T m; V3<T> c; M3<T> I; };
template <class T> Body<T> body_from_params(const double* p) {
    Body<T> b;
    b.m = T(p[0]);
    for (int i = 0; i < 3; ++i) b.c[i] = T(p[1 + i]) / b.m;
    const double* v = p + 4;  // unvech, rigid_body.h:20-25
    double I[9] = {v[0], v[1], v[2], v[1], v[3], v[4], v[2], v[4], v[5]};
    b.I = from_d<T>(I);
    return b;
}

// contact_constraints.h:107-157 + :162-194 (unnormalised), forces length nf*nc
template <class T>
void object_dynamics(const orc_problem* P, const T* forces, const EEState<T>& X, T* out) {
    const int nb = P->nb, nc = P->nc;
    std::vector<V3<T>> F(nb), Tq(nb);
    std::vector<Body<T>> bodies(nb);
    for (int b = 0; b < nb; ++b) {
        bodies[b] = body_from_params<T>(P->body_params[b]);
        F[b] = {{T(0.0), T(0.0), T(0.0)}};
        Tq[b] = F[b];
    }
    const bool frictionless = (P->nf == 1);  // contact_constraints.h:111
    for (int i = 0; i < nc; ++i) {
        V3<T> f;
        if (frictionless) f = scl(from_d3<T>(P->contact_normal[i]), forces[i]);
        else f = {{forces[3 * i], forces[3 * i + 1], forces[3 * i + 2]}};
        int b1 = P->contact_body1[i], b2 = P->contact_body2[i];
        if (b1 >= 0) {  // :126-138
            V3<T> lever = sub(from_d3<T>(P->contact_r1[i]), bodies[b1].c);
            F[b1] = add(F[b1], f);
            Tq[b1] = add(Tq[b1], cross(lever, f));
        }
        {  // :140-154 second object sees the negative force
            V3<T> lever = sub(from_d3<T>(P->contact_r2[i]), bodies[b2].c);
            V3<T> nf_ = scl(f, T(-1.0));
            F[b2] = add(F[b2], nf_);
            Tq[b2] = add(Tq[b2], cross(lever, nf_));
        }
    }
    // util.h:39-44 dC_dtt
    M3<T> Sw = skew3(X.w), Sa = skew3(X.al);
    M3<T> ddC = mm(madd(Sa, mm(Sw, Sw)), X.C);
    V3<T> g0 = from_d3<T>(P->gravity);
    for (int b = 0; b < nb; ++b) {  // contact_constraints.h:80-102
        const Body<T>& B = bodies[b];
        V3<T> acc = sub(add(X.a, mv(ddC, B.c)), g0);
        V3<T> gif = scl(mtv(X.C, acc), B.m);
        V3<T> we = mtv(X.C, X.w), ae = mtv(X.C, X.al);
        V3<T> tau = add(cross(we, mv(B.I, we)), mv(B.I, ae));
        for (int i = 0; i < 3; ++i) {
            out[6 * b + i] = (gif[i] - F[b][i]) / B.m;
            out[6 * b + 3 + i] = (tau[i] - Tq[b][i]) / B.m;
        }
    }
}

// contact_constraints.h:50-77
template <class T>
void friction_rows(const orc_problem* P, const T* forces, T* out) {
    for (int i = 0; i < P->nc; ++i) {
        V3<T> f{{forces[3 * i], forces[3 * i + 1], forces[3 * i + 2]}};
        T fn = dot(from_d3<T>(P->contact_normal[i]), f);
        T t0 = dot(from_d3<T>(P->contact_span[i]), f);
        T t1 = dot(from_d3<T>(P->contact_span[i] + 3), f);
        T mu = T(P->contact_mu[i]);
        out[5 * i + 0] = fn;
        out[5 * i + 1] = mu * fn - t0 - t1;
        out[5 * i + 2] = mu * fn - t0 + t1;
        out[5 * i + 3] = mu * fn + t0 - t1;
        out[5 * i + 4] = mu * fn + t0 + t1;
    }
}

// balancing_constraints.cpp:114-155
template <class T>
void eq_con(const orc_problem* P, const T* x, const T* u, T* g) {
    EEState<T> X = ee_kin<T>(P, x);
    const T* forces = u + P->nq;  // input.tail(dims.f())
    object_dynamics<T>(P, forces, X, g);
    T n = T(std::sqrt(6.0 * P->nb));
    for (int i = 0; i < 6 * P->nb; ++i) g[i] = g[i] / n;
}

// reference_trajectory.h:18-47 (position part) with ocs2 LinearInterpolation::timeSegment
void target_position(const orc_problem* P, double t, double* pd) {
    if (P->n_way <= 1) { for (int i = 0; i < 3; ++i) pd[i] = P->way_p[0][i]; return; }
    int n = P->n_way, idx; double alpha;
    if (t <= P->way_t[0]) { idx = 0; alpha = 1.0; }
    else if (t >= P->way_t[n - 1]) { idx = n - 2; alpha = 0.0; }
    else {
        idx = 0;
        while (idx + 1 < n - 1 && t >= P->way_t[idx + 1]) ++idx;
        alpha = (P->way_t[idx + 1] - t) / (P->way_t[idx + 1] - P->way_t[idx]);
    }
    for (int i = 0; i < 3; ++i) pd[i] = alpha * P->way_p[idx][i] + (1.0 - alpha) * P->way_p[idx + 1][i];
}

// reference_trajectory.h:18-47 (orientation part): q_lhs.slerp(1 - alpha, q_rhs), Eigen's QuaternionBase::slerp (shortest
// arc, linear interpolation of the coefficients where |q_lhs . q_rhs| >= 1 - eps).  Quaternions xyzw.
void target_orientation(const orc_problem* P, double t, double* q) {
    if (P->n_way <= 1) { for (int i = 0; i < 4; ++i) q[i] = P->way_q[0][i]; return; }
    int n = P->n_way, idx; double alpha;
    if (t <= P->way_t[0]) { idx = 0; alpha = 1.0; }
    else if (t >= P->way_t[n - 1]) { idx = n - 2; alpha = 0.0; }
    else {
        idx = 0;
        while (idx + 1 < n - 1 && t >= P->way_t[idx + 1]) ++idx;
        alpha = (P->way_t[idx + 1] - t) / (P->way_t[idx + 1] - P->way_t[idx]);
    }
    const double* a = P->way_q[idx]; const double* b = P->way_q[idx + 1];
    const double tt = 1.0 - alpha;
    const double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3], ad = std::fabs(d);
    double s0, s1;
    if (ad >= 1.0 - 2.220446049250313e-16) { s0 = 1.0 - tt; s1 = tt; }
    else { const double th = std::acos(ad), st = std::sin(th); s0 = std::sin((1.0 - tt) * th) / st; s1 = std::sin(tt * th) / st; }
    if (d < 0) s1 = -s1;
    for (int i = 0; i < 4; ++i) q[i] = s0 * a[i] + s1 * b[i];
}

// Orientation error of the end-effector cost (cost/end_effector_cost.h:42-44; [UPSTREAM] ocs2
// PinocchioEndEffectorKinematicsCppAd::getOrientationError = quaternionDistance(q, q_ref) =
// q_w r_v - r_w q_v + q_v x r_v) with q the quaternion of the end-effector rotation, extracted branch by branch on the
// largest of trace / diagonal entries (Eigen / Pinocchio assignment), in the arithmetic type T.
template <class T>
void orientation_error(const M3<T>& C, const double* r /*xyzw*/, T* e) {
    const T m00 = C(0, 0), m11 = C(1, 1), m22 = C(2, 2);
    const double t = val(m00) + val(m11) + val(m22);
    T q[4];  // xyzw
    if (t > 0) {
        T s = sqrt(m00 + m11 + m22 + T(1.0)) * T(2.0);
        q[3] = s * T(0.25); q[0] = (C(2, 1) - C(1, 2)) / s; q[1] = (C(0, 2) - C(2, 0)) / s; q[2] = (C(1, 0) - C(0, 1)) / s;
    } else if (val(m00) > val(m11) && val(m00) > val(m22)) {
        T s = sqrt(T(1.0) + m00 - m11 - m22) * T(2.0);
        q[3] = (C(2, 1) - C(1, 2)) / s; q[0] = s * T(0.25); q[1] = (C(0, 1) + C(1, 0)) / s; q[2] = (C(0, 2) + C(2, 0)) / s;
    } else if (val(m11) > val(m22)) {
        T s = sqrt(T(1.0) + m11 - m00 - m22) * T(2.0);
        q[3] = (C(0, 2) - C(2, 0)) / s; q[0] = (C(0, 1) + C(1, 0)) / s; q[1] = s * T(0.25); q[2] = (C(1, 2) + C(2, 1)) / s;
    } else {
        T s = sqrt(T(1.0) + m22 - m00 - m11) * T(2.0);
        q[3] = (C(1, 0) - C(0, 1)) / s; q[0] = (C(0, 2) + C(2, 0)) / s; q[1] = (C(1, 2) + C(2, 1)) / s; q[2] = s * T(0.25);
    }
    const T rw = T(r[3]), rx = T(r[0]), ry = T(r[1]), rz = T(r[2]);
    e[0] = q[3] * rx - rw * q[0] + (q[1] * rz - q[2] * ry);
    e[1] = q[3] * ry - rw * q[1] + (q[2] * rx - q[0] * rz);
    e[2] = q[3] * rz - rw * q[2] + (q[0] * ry - q[1] * rx);
}

// ------------------------------------------------------------------------------------------------
// small dense helpers (row-major)
typedef std::vector<double> vec;

bool chol(double* A, int n) {  // in-place lower Cholesky; returns false if not PD
    for (int j = 0; j < n; ++j) {
        double s = A[j * n + j];
        for (int k = 0; k < j; ++k) s -= A[j * n + k] * A[j * n + k];
        if (!(s > 0.0)) return false;
        double d = std::sqrt(s);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double t = A[i * n + j];
            for (int k = 0; k < j; ++k) t -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = t / d;
        }
    }
    return true;
}
// solve L L^T X = B for X (B is n x m row-major, overwritten)
void chol_solve(const double* L, int n, double* B, int m) {
    for (int c = 0; c < m; ++c) {
        for (int i = 0; i < n; ++i) {
            double s = B[i * m + c];
            for (int k = 0; k < i; ++k) s -= L[i * n + k] * B[k * m + c];
            B[i * m + c] = s / L[i * n + i];
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = B[i * m + c];
            for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * B[k * m + c];
            B[i * m + c] = s / L[i * n + i];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dynamics: x = [q, v, a], jerk input; exact discretisation of system_dynamics.h:15-22
// (the reference integrates the same LTI nilpotent system with RK4, controller_interface.cpp:115,
//  which is exact for it)
struct Dyn {
    int nq, nx, nu; double dt;
    void Ax(const double* x, double* y) const {
        double h = dt, h2 = 0.5 * dt * dt;
        for (int i = 0; i < nq; ++i) {
            y[i] = x[i] + h * x[nq + i] + h2 * x[2 * nq + i];
            y[nq + i] = x[nq + i] + h * x[2 * nq + i];
            y[2 * nq + i] = x[2 * nq + i];
        }
    }
    void Bu_add(const double* u, double* y) const {
        double h = dt, h2 = 0.5 * dt * dt, h3 = dt * dt * dt / 6.0;
        for (int i = 0; i < nq; ++i) {
            y[i] += h3 * u[i];
            y[nq + i] += h2 * u[i];
            y[2 * nq + i] += h * u[i];
        }
    }
    void dense(vec& A, vec& B) const {
        A.assign((size_t)nx * nx, 0.0); B.assign((size_t)nx * nu, 0.0);
        double h = dt, h2 = 0.5 * dt * dt, h3 = dt * dt * dt / 6.0;
        for (int i = 0; i < nq; ++i) {
            A[i * nx + i] = 1; A[i * nx + nq + i] = h; A[i * nx + 2 * nq + i] = h2;
            A[(nq + i) * nx + nq + i] = 1; A[(nq + i) * nx + 2 * nq + i] = h;
            A[(2 * nq + i) * nx + 2 * nq + i] = 1;
            B[i * nu + i] = h3; B[(nq + i) * nu + i] = h2; B[(2 * nq + i) * nu + i] = h;
        }
    }
};

// ------------------------------------------------------------------------------------------------
// Stage-wise QP in step variables (dx_k, du_k):
//   min  sum_k 1/2 [dx;du]' [Q 0;0 R] [dx;du] + q'dx + r'du        (R diagonal + nothing else)
//   s.t. dx_{k+1} = A dx_k + B du_k + b_k
//        Ce_k dx_k + De_k du_k + e_k = 0                            (k < N)
//        CN dx_N + eN = 0
//        box:  xl_k <= dx_k <= xu_k (k >= 1),  ul_k <= du_k <= uu_k
//        poly: Gu_k du_k + d_k >= 0                                 (friction rows)
struct StageQP {
    vec Q, q, Rd, r, b;      // Q nx*nx, Rd diag nu
    vec Ce, De, e;           // ne rows
    vec xl, xu, ul, uu;      // box on the step
    vec Gu, d;               // np rows on du
    vec Gx, gd;              // no rows on dx: Gx dx + gd >= 0 (obstacle rows, stages 1..N-1)
};
struct QP {
    int N, nx, nu, ne, np, neN, nfc, no = 0;
    // soft constraints (ocs2 hpipm_interface SlackSettings, upright_control/src/pybindings.cpp:160-181; defaults of
    // wrappers.py:121-143): which row classes get a slack sigma >= 0 with cost 1/2 Z sigma^2 + z sigma
    bool soft_x = false, soft_u = false, soft_poly = false, soft_eq = false;
    double ZL = 100, ZU = 100, zL = 0, zU = 0;
    vec A, B;
    std::vector<StageQP> st;  // N stages
    vec QN, qN, CN, eN, xlN, xuN;
    vec dx0;
};

struct QPSol {
    std::vector<vec> dx, du, pi, nu_;  // dx[0..N], du[0..N-1], pi[0..N], nu_[0..N-1]
    std::vector<vec> K;                // feedback gains of the last factorisation, ocs2 sign (du = K ddx + ...), [N][nu*nx]
    vec yN;
    int iters, status;
    double res[4];
};

// Riccati factorisation of the barrier-augmented LQ problem with stage equalities handled by a
// Schur complement (exact) and the terminal equality by a proximal penalty (rhoN) whose multiplier
// is carried explicitly so that the fixed point is the exact KKT point.
struct Riccati {
    const QP* qp;
    double rho_s, rhoN;
    std::vector<vec> P, Lr, Kx, Y, Ls, Cbar;  // per stage
    // workspace of the vector pass
    std::vector<vec> p, ku0, snu;

    bool factor(const std::vector<vec>& Hxx_add /*diag barrier on x, per k (0..N)*/,
                const std::vector<vec>& Huu_add /*full nu*nu barrier on u per k*/,
                const std::vector<vec>& Hxx_dense /*full nx*nx barrier on x per k (may be empty)*/) {
        const int N = qp->N, nx = qp->nx, nu = qp->nu, ne = qp->ne;
        P.assign(N + 1, vec()); Lr.assign(N, vec()); Kx.assign(N, vec()); Y.assign(N, vec());
        Ls.assign(N, vec()); Cbar.assign(N, vec());
        // terminal
        P[N] = qp->QN;
        for (int i = 0; i < nx; ++i) P[N][i * nx + i] += Hxx_add[N][i];
        for (int r = 0; r < qp->neN; ++r)
            for (int i = 0; i < nx; ++i)
                for (int j = 0; j < nx; ++j) P[N][i * nx + j] += qp->CN[r * nx + i] * qp->CN[r * nx + j] / rhoN;
        vec W((size_t)nx * nx), Hxx((size_t)nx * nx), Hux((size_t)nu * nx), Huu((size_t)nu * nu);
        const vec& A = qp->A; const vec& B = qp->B;
        for (int k = N - 1; k >= 0; --k) {
            const StageQP& s = qp->st[k];
            // W = P+ A
            for (int i = 0; i < nx; ++i)
                for (int j = 0; j < nx; ++j) {
                    double t = 0; for (int l = 0; l < nx; ++l) t += P[k + 1][i * nx + l] * A[l * nx + j];
                    W[i * nx + j] = t;
                }
            for (int i = 0; i < nx; ++i)
                for (int j = 0; j < nx; ++j) {
                    double t = s.Q[i * nx + j]; for (int l = 0; l < nx; ++l) t += A[l * nx + i] * W[l * nx + j];
                    Hxx[i * nx + j] = t;
                }
            for (int i = 0; i < nx; ++i) Hxx[i * nx + i] += Hxx_add[k][i];
            if (!Hxx_dense.empty() && !Hxx_dense[k].empty()) for (size_t i = 0; i < Hxx.size(); ++i) Hxx[i] += Hxx_dense[k][i];
            for (int i = 0; i < nu; ++i)
                for (int j = 0; j < nx; ++j) {
                    double t = 0; for (int l = 0; l < nx; ++l) t += B[l * nu + i] * W[l * nx + j];
                    Hux[i * nx + j] = t;
                }
            // Huu = R + barrier + B' P+ B
            vec PB((size_t)nx * nu);
            for (int i = 0; i < nx; ++i)
                for (int j = 0; j < nu; ++j) {
                    double t = 0; for (int l = 0; l < nx; ++l) t += P[k + 1][i * nx + l] * B[l * nu + j];
                    PB[i * nu + j] = t;
                }
            for (int i = 0; i < nu; ++i)
                for (int j = 0; j < nu; ++j) {
                    double t = Huu_add[k][i * nu + j] + (i == j ? s.Rd[i] : 0.0);
                    for (int l = 0; l < nx; ++l) t += B[l * nu + i] * PB[l * nu + j];
                    Huu[i * nu + j] = t;
                }
            Lr[k] = Huu;
            if (!chol(Lr[k].data(), nu)) return false;
            Kx[k] = Hux; chol_solve(Lr[k].data(), nu, Kx[k].data(), nx);  // Huu^-1 Hux
            // P = Hxx - Hux' Kx
            P[k].assign((size_t)nx * nx, 0.0);
            for (int i = 0; i < nx; ++i)
                for (int j = 0; j < nx; ++j) {
                    double t = Hxx[i * nx + j]; for (int l = 0; l < nu; ++l) t -= Hux[l * nx + i] * Kx[k][l * nx + j];
                    P[k][i * nx + j] = t;
                }
            if (ne > 0) {
                // Y = Huu^-1 De'  (nu x ne)
                Y[k].assign((size_t)nu * ne, 0.0);
                for (int i = 0; i < nu; ++i) for (int r = 0; r < ne; ++r) Y[k][i * ne + r] = s.De[r * nu + i];
                chol_solve(Lr[k].data(), nu, Y[k].data(), ne);
                Ls[k].assign((size_t)ne * ne, 0.0);
                for (int r = 0; r < ne; ++r)
                    for (int c = 0; c < ne; ++c) {
                        double t = (r == c ? rho_s : 0.0); for (int l = 0; l < nu; ++l) t += s.De[r * nu + l] * Y[k][l * ne + c];
                        Ls[k][r * ne + c] = t;
                    }
                if (!chol(Ls[k].data(), ne)) return false;
                // Cbar = Ce - De Kx
                Cbar[k].assign((size_t)ne * nx, 0.0);
                for (int r = 0; r < ne; ++r)
                    for (int j = 0; j < nx; ++j) {
                        double t = s.Ce[r * nx + j]; for (int l = 0; l < nu; ++l) t -= s.De[r * nu + l] * Kx[k][l * nx + j];
                        Cbar[k][r * nx + j] = t;
                    }
                vec SC = Cbar[k]; chol_solve(Ls[k].data(), ne, SC.data(), nx);  // S^-1 Cbar
                for (int i = 0; i < nx; ++i)
                    for (int j = 0; j < nx; ++j) {
                        double t = 0; for (int r = 0; r < ne; ++r) t += Cbar[k][r * nx + i] * SC[r * nx + j];
                        P[k][i * nx + j] += t;
                    }
            }
            // symmetrise
            for (int i = 0; i < nx; ++i) for (int j = i + 1; j < nx; ++j) {
                double t = 0.5 * (P[k][i * nx + j] + P[k][j * nx + i]); P[k][i * nx + j] = P[k][j * nx + i] = t;
            }
        }
        return true;
    }

    // solve for the step given gradients gx[k] (k=0..N), gu[k], defects b[k], eq residuals e[k], eN.
    // outputs dx (dx[0] given), du, pi (costates, k=0..N), nu (stage eq multipliers), dyN.
    void solve(const std::vector<vec>& gx, const std::vector<vec>& gu, const std::vector<vec>& b,
               const std::vector<vec>& e, const vec& eN, std::vector<vec>& dx, std::vector<vec>& du,
               std::vector<vec>& pi, std::vector<vec>& nu_, vec& dyN) {
        const int N = qp->N, nx = qp->nx, nu = qp->nu, ne = qp->ne;
        const vec& A = qp->A; const vec& B = qp->B;
        p.assign(N + 1, vec(nx)); ku0.assign(N, vec(nu)); snu.assign(N, vec(ne));
        for (int i = 0; i < nx; ++i) {
            double t = gx[N][i];
            for (int r = 0; r < qp->neN; ++r) t += qp->CN[r * nx + i] * eN[r] / rhoN;
            p[N][i] = t;
        }
        vec w(nx), hx(nx), hu(nu), ee(ne);
        for (int k = N - 1; k >= 0; --k) {
            const StageQP& s = qp->st[k];
            for (int i = 0; i < nx; ++i) {
                double t = p[k + 1][i]; for (int l = 0; l < nx; ++l) t += P[k + 1][i * nx + l] * b[k][l];
                w[i] = t;
            }
            for (int i = 0; i < nx; ++i) { double t = gx[k][i]; for (int l = 0; l < nx; ++l) t += A[l * nx + i] * w[l]; hx[i] = t; }
            for (int i = 0; i < nu; ++i) { double t = gu[k][i]; for (int l = 0; l < nx; ++l) t += B[l * nu + i] * w[l]; hu[i] = t; }
            ku0[k] = hu; chol_solve(Lr[k].data(), nu, ku0[k].data(), 1);
            for (int i = 0; i < nx; ++i) { double t = hx[i]; for (int l = 0; l < nu; ++l) t -= Kx[k][l * nx + i] * hu[l]; p[k][i] = t; }
            if (ne > 0) {
                for (int r = 0; r < ne; ++r) { double t = e[k][r]; for (int l = 0; l < nu; ++l) t -= s.De[r * nu + l] * ku0[k][l]; ee[r] = t; }
                snu[k] = ee; chol_solve(Ls[k].data(), ne, snu[k].data(), 1);
                for (int i = 0; i < nx; ++i) { double t = 0; for (int r = 0; r < ne; ++r) t += Cbar[k][r * nx + i] * snu[k][r]; p[k][i] += t; }
            }
        }
        // forward
        for (int i = 0; i < nx; ++i) { double t = p[0][i]; for (int l = 0; l < nx; ++l) t += P[0][i * nx + l] * dx[0][l]; pi[0][i] = t; }
        for (int k = 0; k < N; ++k) {
            if (ne > 0) {
                vec rhs(ne);
                for (int r = 0; r < ne; ++r) { double t = 0; for (int j = 0; j < nx; ++j) t += Cbar[k][r * nx + j] * dx[k][j]; rhs[r] = t; }
                chol_solve(Ls[k].data(), ne, rhs.data(), 1);
                for (int r = 0; r < ne; ++r) nu_[k][r] = rhs[r] + snu[k][r];
            }
            for (int i = 0; i < nu; ++i) {
                double t = -ku0[k][i];
                for (int j = 0; j < nx; ++j) t -= Kx[k][i * nx + j] * dx[k][j];
                for (int r = 0; r < ne; ++r) t -= Y[k][i * ne + r] * nu_[k][r];
                du[k][i] = t;
            }
            for (int i = 0; i < nx; ++i) {
                double t = b[k][i];
                for (int j = 0; j < nx; ++j) t += A[i * nx + j] * dx[k][j];
                for (int j = 0; j < nu; ++j) t += B[i * nu + j] * du[k][j];
                dx[k + 1][i] = t;
            }
            for (int i = 0; i < nx; ++i) {
                double t = p[k + 1][i]; for (int l = 0; l < nx; ++l) t += P[k + 1][i * nx + l] * dx[k + 1][l];
                pi[k + 1][i] = t;
            }
        }
        for (int r = 0; r < qp->neN; ++r) {
            double t = eN[r]; for (int j = 0; j < nx; ++j) t += qp->CN[r * nx + j] * dx[N][j];
            dyN[r] = t / rhoN;
        }
    }
};

// Mehrotra predictor-corrector primal-dual IPM (published algorithm; HPIPM [UPSTREAM] is the
// reference's implementation of the same family, settings at upright_control/src/pybindings.cpp:183-188).
// Unknown: z = (dx, du) with dx[0] fixed.  Inequalities c_i(z) >= 0 with slack t and dual lam.
int ipm_solve(const QP& qp, int iter_max, double tol, double tol_stat, QPSol& sol) {
    if (!(tol_stat > 0.0)) tol_stat = tol;
    const int N = qp.N, nx = qp.nx, nu = qp.nu, ne = qp.ne, np = qp.np, neN = qp.neN;
    // inequality layout per stage k<N: [x lower nx][x upper nx] (k>=1 only) [u lower nu][u upper nu][poly np]
    // terminal: [x lower][x upper]
    // state-polytopic (obstacle) rows sit behind the others at stages 1..N-1
    const int no = qp.no;
    auto has_obs = [&](int k) { return no > 0 && k >= 1 && k < N; };
    const int oo = 2 * nx + 2 * nu + np;
    auto nik = [&](int k) { return (k >= 1 ? 2 * nx : 0) + (k < N ? 2 * nu + np : 0) + (has_obs(k) ? no : 0); };
    // row class of slot i of stage k: 0 x lower, 1 x upper, 2 u lower, 3 u upper, 4 polytopic (friction / collision / projectile)
    auto row_class = [&](int k, int i) {
        if (k >= 1) { if (i < nx) return 0; if (i < 2 * nx) return 1; i -= 2 * nx; }
        if (k < N) { if (i < nu) return 2; if (i < 2 * nu) return 3; }
        return 4;
    };
    auto is_soft = [&](int k, int i) { const int c = row_class(k, i); return c < 2 ? qp.soft_x : (c < 4 ? qp.soft_u : qp.soft_poly); };
    auto pen_Z = [&](int k, int i) { const int c = row_class(k, i); return (c == 1 || c == 3) ? qp.ZU : qp.ZL; };
    auto pen_z = [&](int k, int i) { const int c = row_class(k, i); return (c == 1 || c == 3) ? qp.zU : qp.zL; };
    const bool any_soft = qp.soft_x || qp.soft_u || qp.soft_poly;
    std::vector<vec> t(N + 1), lam(N + 1), dt_(N + 1), dlam(N + 1), dt_aff(N + 1), dlam_aff(N + 1), cval(N + 1);
    std::vector<vec> dx(N + 1, vec(nx, 0.0)), du(N, vec(nu, 0.0)), pi(N + 1, vec(nx, 0.0)), nu_(N, vec(ne, 0.0));
    std::vector<vec> ddx(N + 1, vec(nx, 0.0)), ddu(N, vec(nu, 0.0)), pi_new(N + 1, vec(nx, 0.0)), nu_new(N, vec(ne, 0.0));
    vec yN(neN, 0.0), dyN(neN, 0.0);
    dx[0] = qp.dx0;
    int ntot = 0;
    for (int k = 0; k <= N; ++k) { int n = nik(k); t[k].assign(n, 0); lam[k].assign(n, 0); dt_[k].assign(n, 0); dlam[k].assign(n, 0); dt_aff[k] = dt_[k]; dlam_aff[k] = dt_[k]; cval[k] = dt_[k]; ntot += n; }
    // slack part of the softened rows: sigma >= 0 (tau its own slack, gam its multiplier)
    std::vector<vec> sig(N + 1), tau(N + 1), gam(N + 1), dsig(N + 1), dtau(N + 1), dgam(N + 1), dtau_aff(N + 1), dgam_aff(N + 1), rcs(N + 1), rps(N + 1);
    std::vector<vec> weff(N + 1), geff(N + 1);
    int nsoft = 0;
    for (int k = 0; k <= N; ++k) {
        const size_t n = t[k].size();
        for (auto* v : {&sig[k], &tau[k], &gam[k], &dsig[k], &dtau[k], &dgam[k], &dtau_aff[k], &dgam_aff[k], &rcs[k], &rps[k], &weff[k], &geff[k]}) v->assign(n, 0.0);
        for (size_t i = 0; i < n; ++i) if (is_soft(k, (int)i)) ++nsoft;
    }
    ntot += nsoft;

    // c(z) for each inequality
    auto eval_c = [&](const std::vector<vec>& X, const std::vector<vec>& U, std::vector<vec>& c) {
        for (int k = 0; k <= N; ++k) {
            int o = 0;
            if (k >= 1) {
                const vec& xl = (k < N ? qp.st[k].xl : qp.xlN); const vec& xu = (k < N ? qp.st[k].xu : qp.xuN);
                for (int i = 0; i < nx; ++i) c[k][o + i] = X[k][i] - xl[i];
                for (int i = 0; i < nx; ++i) c[k][o + nx + i] = xu[i] - X[k][i];
                o += 2 * nx;
            }
            if (k < N) {
                const StageQP& s = qp.st[k];
                for (int i = 0; i < nu; ++i) c[k][o + i] = U[k][i] - s.ul[i];
                for (int i = 0; i < nu; ++i) c[k][o + nu + i] = s.uu[i] - U[k][i];
                o += 2 * nu;
                for (int r = 0; r < np; ++r) { double v = s.d[r]; for (int j = 0; j < nu; ++j) v += s.Gu[r * nu + j] * U[k][j]; c[k][o + r] = v; }
            }
            if (has_obs(k)) {
                const StageQP& s = qp.st[k];
                for (int r = 0; r < no; ++r) { double v = s.gd[r]; for (int j = 0; j < nx; ++j) v += s.Gx[r * nx + j] * X[k][j]; c[k][oo + r] = v; }
            }
        }
    };
    // initial point: z = 0 (dx0 fixed), t = max(c, thr), lam = mu0 / t
    eval_c(dx, du, cval);
    double thr = 3.0, mu0 = 0.1;   // on the plateau of the iteration count over (thr, mu0) for every BASELINE configuration
    if (getenv("ORC_THR")) thr = atof(getenv("ORC_THR"));
    if (getenv("ORC_MU0")) mu0 = atof(getenv("ORC_MU0"));
    for (int k = 0; k <= N; ++k)
        for (size_t i = 0; i < t[k].size(); ++i) { t[k][i] = std::max(cval[k][i], thr); lam[k][i] = mu0 / t[k][i]; }
    if (any_soft) for (int k = 0; k <= N; ++k)
        for (size_t i = 0; i < t[k].size(); ++i) if (is_soft(k, (int)i)) { sig[k][i] = 0.0; tau[k][i] = thr; gam[k][i] = mu0 / thr; }

    // forces are the tail of u: the force block spans the equality rows iff nu - nq >= ne
    // softened stage equality (soft_eq): the row's slack pair with equal L2 penalties Z and no L1 term eliminates to the
    // penalty Z/2 |C dz + e|^2 = the regularised equality C dz + e = nu / Z: rho_s = 1 / Z, no proximal carry-over
    const double rho_soft = qp.soft_eq ? 1.0 / qp.ZL : 0.0;
    const double rho_prox = qp.soft_eq ? 0.0 : ((qp.nfc < ne) ? 1e-6 : 0.0);
    Riccati ric; ric.qp = &qp; ric.rho_s = qp.soft_eq ? rho_soft : ((qp.nfc < ne) ? 1e-6 : 1e-12); ric.rhoN = 1e-6;
    std::vector<vec> Hxx_add(N + 1, vec(nx, 0.0)), Huu_add(N, vec((size_t)nu * nu, 0.0)), Hxx_dense(no > 0 ? N + 1 : 0);
    std::vector<vec> gx(N + 1, vec(nx)), gu(N, vec(nu)), bres(N, vec(nx)), eres(N, vec(ne));
    std::vector<vec> rp(N + 1), rc(N + 1);
    for (int k = 0; k <= N; ++k) { rp[k].assign(t[k].size(), 0); rc[k] = rp[k]; }
    vec eNres(neN, 0.0);
    const vec& A = qp.A; const vec& B = qp.B;
    sol.status = 1;
    int it = 0;
    for (; it <= iter_max; ++it) {
        // ---------------- residuals at the current iterate
        eval_c(dx, du, cval);
        double mu = 0, r_ineq = 0;
        for (int k = 0; k <= N; ++k) for (size_t i = 0; i < t[k].size(); ++i) {
            rp[k][i] = cval[k][i] + sig[k][i] - t[k][i]; r_ineq = std::max(r_ineq, std::fabs(rp[k][i])); mu += lam[k][i] * t[k][i];
            if (any_soft && is_soft(k, (int)i)) { rps[k][i] = sig[k][i] - tau[k][i]; r_ineq = std::max(r_ineq, std::fabs(rps[k][i])); mu += gam[k][i] * tau[k][i]; }
        }
        mu /= std::max(1, ntot);
        double r_eq = 0, r_stat = 0;
        for (int k = 0; k < N; ++k) {
            const StageQP& s = qp.st[k];
            for (int i = 0; i < nx; ++i) {
                double v = s.b[i] - dx[k + 1][i];
                for (int j = 0; j < nx; ++j) v += A[i * nx + j] * dx[k][j];
                for (int j = 0; j < nu; ++j) v += B[i * nu + j] * du[k][j];
                bres[k][i] = v; r_eq = std::max(r_eq, std::fabs(v));
            }
            for (int r = 0; r < ne; ++r) {
                double v = s.e[r];
                for (int j = 0; j < nx; ++j) v += s.Ce[r * nx + j] * dx[k][j];
                for (int j = 0; j < nu; ++j) v += s.De[r * nu + j] * du[k][j];
                eres[k][r] = v; r_eq = std::max(r_eq, std::fabs(v - rho_soft * nu_[k][r]));
            }
        }
        for (int r = 0; r < neN; ++r) {
            double v = qp.eN[r]; for (int j = 0; j < nx; ++j) v += qp.CN[r * nx + j] * dx[N][j];
            eNres[r] = v; r_eq = std::max(r_eq, std::fabs(v));
        }
        // cost gradient minus G' lam  (the part of stationarity that does not involve pi, nu)
        for (int k = 0; k <= N; ++k) {
            int o = 0;
            const vec& Q = (k < N ? qp.st[k].Q : qp.QN); const vec& q = (k < N ? qp.st[k].q : qp.qN);
            for (int i = 0; i < nx; ++i) { double v = q[i]; for (int j = 0; j < nx; ++j) v += Q[i * nx + j] * dx[k][j]; gx[k][i] = v; }
            if (k >= 1) { for (int i = 0; i < nx; ++i) gx[k][i] += -lam[k][i] + lam[k][nx + i]; o += 2 * nx; }
            if (k < N) {
                const StageQP& s = qp.st[k];
                for (int i = 0; i < nu; ++i) gu[k][i] = s.Rd[i] * du[k][i] + s.r[i] - lam[k][o + i] + lam[k][o + nu + i];
                for (int r = 0; r < np; ++r) for (int j = 0; j < nu; ++j) gu[k][j] -= s.Gu[r * nu + j] * lam[k][o + 2 * nu + r];
            }
            if (has_obs(k)) for (int r = 0; r < no; ++r) for (int j = 0; j < nx; ++j) gx[k][j] -= qp.st[k].Gx[r * nx + j] * lam[k][oo + r];
        }
        // full stationarity residual with current multipliers
        for (int k = 0; k <= N; ++k) {
            if (k >= 1) for (int i = 0; i < nx; ++i) {
                double v = gx[k][i] - pi[k][i];
                if (k < N) { for (int l = 0; l < nx; ++l) v += A[l * nx + i] * pi[k + 1][l]; for (int r = 0; r < ne; ++r) v += qp.st[k].Ce[r * nx + i] * nu_[k][r]; }
                else for (int r = 0; r < neN; ++r) v += qp.CN[r * nx + i] * yN[r];
                r_stat = std::max(r_stat, std::fabs(v));
            }
            if (k < N) for (int i = 0; i < nu; ++i) {
                double v = gu[k][i];
                for (int l = 0; l < nx; ++l) v += B[l * nu + i] * pi[k + 1][l];
                for (int r = 0; r < ne; ++r) v += qp.st[k].De[r * nu + i] * nu_[k][r];
                r_stat = std::max(r_stat, std::fabs(v));
            }
        }
        if (any_soft) for (int k = 0; k <= N; ++k) for (size_t i = 0; i < t[k].size(); ++i) if (is_soft(k, (int)i))
            r_stat = std::max(r_stat, std::fabs(pen_Z(k, (int)i) * sig[k][i] + pen_z(k, (int)i) - lam[k][i] - gam[k][i]));
                sol.res[0] = r_stat; sol.res[1] = r_eq; sol.res[2] = r_ineq; sol.res[3] = mu;
        if (getenv("ORC_DEBUG")) printf("orc it %d res %.3e %.3e %.3e %.3e\n", it, r_stat, r_eq, r_ineq, mu);
        if (it > 0 && r_stat < tol_stat && r_eq < tol && r_ineq < tol && mu < tol) { sol.status = 0; break; }
        if (it == iter_max) break;

        // effective barrier weight of every row: lam / t, and for a softened row the same after the elimination of its
        // slack, w (Z + w_s) / (Z + w + w_s) with w_s = gam / tau
        for (int k = 0; k <= N; ++k) for (size_t i = 0; i < t[k].size(); ++i) {
            const double w = lam[k][i] / t[k][i];
            weff[k][i] = w;
            if (any_soft && is_soft(k, (int)i)) { const double Z = pen_Z(k, (int)i), ws = gam[k][i] / tau[k][i]; weff[k][i] = w * (Z + ws) / (Z + w + ws); }
        }
        // ---------------- factorisation with W = lam / t
        for (int k = 0; k <= N; ++k) {
            int o = 0;
            std::fill(Hxx_add[k].begin(), Hxx_add[k].end(), 0.0);
            if (k >= 1) { for (int i = 0; i < nx; ++i) Hxx_add[k][i] = weff[k][i] + weff[k][nx + i]; o += 2 * nx; }
            if (k < N) {
                const StageQP& s = qp.st[k];
                std::fill(Huu_add[k].begin(), Huu_add[k].end(), 0.0);
                for (int i = 0; i < nu; ++i) Huu_add[k][i * nu + i] = weff[k][o + i] + weff[k][o + nu + i];
                for (int r = 0; r < np; ++r) {
                    double w = weff[k][o + 2 * nu + r];
                    for (int i = 0; i < nu; ++i) { double gi = s.Gu[r * nu + i]; if (gi == 0.0) continue;
                        for (int j = 0; j < nu; ++j) Huu_add[k][i * nu + j] += w * gi * s.Gu[r * nu + j]; }
                }
            }
        }
        for (int k = 0; k <= N && no > 0; ++k) {
            Hxx_dense[k].clear();
            if (!has_obs(k)) continue;
            Hxx_dense[k].assign((size_t)nx * nx, 0.0);
            const StageQP& s = qp.st[k];
            for (int r = 0; r < no; ++r) {
                const double w = weff[k][oo + r];
                for (int i = 0; i < nx; ++i) { const double gi = s.Gx[r * nx + i]; if (gi == 0.0) continue;
                    for (int j = 0; j < nx; ++j) Hxx_dense[k][i * nx + j] += w * gi * s.Gx[r * nx + j]; }
            }
        }
        if (!ric.factor(Hxx_add, Huu_add, Hxx_dense)) { sol.status = 2; break; }

        // solve the Newton system for a given complementarity target rc (per inequality):
        //   lam*dt + t*dlam = -rc ;  dt = G dz + rp ;  =>  dlam = -(rc + lam*(G dz + rp)) / t
        auto newton = [&](std::vector<vec>& dT, std::vector<vec>& dL, std::vector<vec>& dTau, std::vector<vec>& dGam) {
            std::vector<vec> hx = gx, hu = gu;
            // gradient multiplier of every row, (rc + lam rp) / t; a softened row also carries what the elimination of its
            // slack leaves behind: - w a / D, a = r_sigma + (rc + lam rp) / t + (rc_s + gam rp_s) / tau, D = Z + w + w_s
            for (int k = 0; k <= N; ++k) for (size_t i = 0; i < t[k].size(); ++i) {
                const double g0 = (rc[k][i] + lam[k][i] * rp[k][i]) / t[k][i];
                geff[k][i] = g0;
                if (any_soft && is_soft(k, (int)i)) {
                    const double Z = pen_Z(k, (int)i), w = lam[k][i] / t[k][i], ws = gam[k][i] / tau[k][i];
                    const double a = (Z * sig[k][i] + pen_z(k, (int)i) - lam[k][i] - gam[k][i]) + g0 + (rcs[k][i] + gam[k][i] * rps[k][i]) / tau[k][i];
                    geff[k][i] = g0 - w * a / (Z + w + ws);
                }
            }
            for (int k = 0; k <= N; ++k) {
                int o = 0;
                // gradient of the reduced system: g - G' * ( -(rc + lam*rp)/t )  = g + G'(rc + lam rp)/t
                if (k >= 1) {
                    for (int i = 0; i < nx; ++i) {
                        hx[k][i] += geff[k][i];
                        hx[k][i] -= geff[k][nx + i];
                    }
                    o += 2 * nx;
                }
                if (k < N) {
                    const StageQP& s = qp.st[k];
                    for (int i = 0; i < nu; ++i) {
                        hu[k][i] += geff[k][o + i];
                        hu[k][i] -= geff[k][o + nu + i];
                    }
                    for (int r = 0; r < np; ++r) {
                        int ii = o + 2 * nu + r; double w = geff[k][ii];
                        for (int j = 0; j < nu; ++j) hu[k][j] += s.Gu[r * nu + j] * w;
                    }
                }
                if (has_obs(k)) for (int r = 0; r < no; ++r) {
                    const int ii = oo + r; const double w = geff[k][ii];
                    for (int j = 0; j < nx; ++j) hx[k][j] += qp.st[k].Gx[r * nx + j] * w;
                }
            }
            // terminal multiplier enters the terminal gradient through the proximal term (true residual)
            std::vector<vec> hxN = hx;
            for (int i = 0; i < nx; ++i) for (int r = 0; r < neN; ++r) hxN[N][i] += qp.CN[r * nx + i] * yN[r];
            ddx[0].assign(nx, 0.0);  // dx[0] is fixed -> zero step
            // rank-deficient force block (frictionless arrangements): proximal form of the stage equality,
            // C dz + e = rho_s (nu+ - nu); the IPM iterations double as the proximal iterations
            std::vector<vec> eprox = eres;
            if (rho_prox > 0.0) for (int k = 0; k < N; ++k) for (int r = 0; r < ne; ++r) eprox[k][r] += rho_prox * nu_[k][r];
            ric.solve(hxN, hu, bres, eprox, eNres, ddx, ddu, pi_new, nu_new, dyN);
            // Iterative refinement of the Newton step (HPIPM: itref): the residual of the reduced KKT system at the computed
            // step, solved for a correction with the factorisation at hand.  Only when the residual is far above rounding
            // (> 1e-8): with barrier weights of 1e8 and more on several rows of one contact the dense factorisation of Huu
            // loses the digits the stationarity test (1e-6) needs, and the iteration then sits on that floor until the cap --
            // two stacked 20 g dice, a projectile row active at a converged plan.  Well-conditioned steps are left bit for bit.
            for (int ref = 0; ref < 3; ++ref) {
                std::vector<vec> r1(N + 1, vec(nx, 0.0)), r2(N, vec(nu, 0.0)), r3(N, vec(nx, 0.0)), r4(N, vec(ne, 0.0));
                double rmax = 0.0;
                for (int k = 1; k <= N; ++k) for (int i = 0; i < nx; ++i) {
                    double v = hxN[k][i] - pi_new[k][i] + Hxx_add[k][i] * ddx[k][i];
                    const vec& Q = (k < N ? qp.st[k].Q : qp.QN);
                    for (int j = 0; j < nx; ++j) v += Q[i * nx + j] * ddx[k][j];
                    if (k < N) {
                        if (!Hxx_dense.empty() && !Hxx_dense[k].empty()) for (int j = 0; j < nx; ++j) v += Hxx_dense[k][i * nx + j] * ddx[k][j];
                        for (int l = 0; l < nx; ++l) v += A[l * nx + i] * pi_new[k + 1][l];
                        for (int r = 0; r < ne; ++r) v += qp.st[k].Ce[r * nx + i] * nu_new[k][r];
                    } else for (int r = 0; r < neN; ++r) {
                        double c = eNres[r]; for (int j = 0; j < nx; ++j) c += qp.CN[r * nx + j] * ddx[N][j];
                        v += qp.CN[r * nx + i] * c / ric.rhoN;
                    }
                    r1[k][i] = v; rmax = std::max(rmax, std::fabs(v));
                }
                for (int k = 0; k < N; ++k) {
                    const StageQP& s = qp.st[k];
                    for (int i = 0; i < nu; ++i) {
                        double v = hu[k][i] + s.Rd[i] * ddu[k][i];
                        for (int j = 0; j < nu; ++j) v += Huu_add[k][i * nu + j] * ddu[k][j];
                        for (int l = 0; l < nx; ++l) v += B[l * nu + i] * pi_new[k + 1][l];
                        for (int r = 0; r < ne; ++r) v += s.De[r * nu + i] * nu_new[k][r];
                        r2[k][i] = v; rmax = std::max(rmax, std::fabs(v));
                    }
                    for (int i = 0; i < nx; ++i) {
                        double v = bres[k][i] - ddx[k + 1][i];
                        for (int j = 0; j < nx; ++j) v += A[i * nx + j] * ddx[k][j];
                        for (int j = 0; j < nu; ++j) v += B[i * nu + j] * ddu[k][j];
                        r3[k][i] = v; rmax = std::max(rmax, std::fabs(v));
                    }
                    for (int r = 0; r < ne; ++r) {
                        double v = eprox[k][r] - ric.rho_s * nu_new[k][r];
                        for (int j = 0; j < nx; ++j) v += s.Ce[r * nx + j] * ddx[k][j];
                        for (int j = 0; j < nu; ++j) v += s.De[r * nu + j] * ddu[k][j];
                        r4[k][r] = v; rmax = std::max(rmax, std::fabs(v));
                    }
                }
                if (!(rmax > 1e-8)) break;
                std::vector<vec> cx(N + 1, vec(nx, 0.0)), cu(N, vec(nu, 0.0)), cpi(N + 1, vec(nx, 0.0)), cnu(N, vec(ne, 0.0));
                vec cyN(neN, 0.0), zN(neN, 0.0);
                ric.solve(r1, r2, r3, r4, zN, cx, cu, cpi, cnu, cyN);
                for (int k = 0; k <= N; ++k) for (int i = 0; i < nx; ++i) { if (k >= 1) ddx[k][i] += cx[k][i]; pi_new[k][i] += cpi[k][i]; }
                for (int k = 0; k < N; ++k) { for (int i = 0; i < nu; ++i) ddu[k][i] += cu[k][i]; for (int r = 0; r < ne; ++r) nu_new[k][r] += cnu[k][r]; }
                for (int r = 0; r < neN; ++r) dyN[r] += cyN[r];
            }
            for (int k = 0; k <= N; ++k) {
                int o = 0;
                if (k >= 1) {
                    for (int i = 0; i < nx; ++i) { dT[k][i] = ddx[k][i] + rp[k][i]; dT[k][nx + i] = -ddx[k][i] + rp[k][nx + i]; }
                    o += 2 * nx;
                }
                if (k < N) {
                    const StageQP& s = qp.st[k];
                    for (int i = 0; i < nu; ++i) { dT[k][o + i] = ddu[k][i] + rp[k][o + i]; dT[k][o + nu + i] = -ddu[k][i] + rp[k][o + nu + i]; }
                    for (int r = 0; r < np; ++r) { double v = rp[k][o + 2 * nu + r]; for (int j = 0; j < nu; ++j) v += s.Gu[r * nu + j] * ddu[k][j]; dT[k][o + 2 * nu + r] = v; }
                }
                if (has_obs(k)) for (int r = 0; r < no; ++r) { double v = rp[k][oo + r]; for (int j = 0; j < nx; ++j) v += qp.st[k].Gx[r * nx + j] * ddx[k][j]; dT[k][oo + r] = v; }
                if (any_soft) for (size_t i = 0; i < t[k].size(); ++i) if (is_soft(k, (int)i)) {
                    // dT so far is G dz + rp; the slack step follows from the eliminated row of the Newton system
                    const double Z = pen_Z(k, (int)i), w = lam[k][i] / t[k][i], ws = gam[k][i] / tau[k][i];
                    const double gdz = dT[k][i] - rp[k][i];
                    const double a = (Z * sig[k][i] + pen_z(k, (int)i) - lam[k][i] - gam[k][i]) + (rc[k][i] + lam[k][i] * rp[k][i]) / t[k][i]
                                     + (rcs[k][i] + gam[k][i] * rps[k][i]) / tau[k][i];
                    dsig[k][i] = -(a + w * gdz) / (Z + w + ws);
                    dT[k][i] += dsig[k][i];
                    dTau[k][i] = dsig[k][i] + rps[k][i];
                    dGam[k][i] = -(rcs[k][i] + gam[k][i] * dTau[k][i]) / tau[k][i];
                }
                for (size_t i = 0; i < t[k].size(); ++i) dL[k][i] = -(rc[k][i] + lam[k][i] * dT[k][i]) / t[k][i];
            }
        };
        auto max_step = [&](const std::vector<vec>& dT, const std::vector<vec>& dL, const std::vector<vec>& dTau, const std::vector<vec>& dGam) {
            double a = 1.0;
            for (int k = 0; k <= N; ++k) for (size_t i = 0; i < t[k].size(); ++i) {
                if (dT[k][i] < 0) a = std::min(a, -t[k][i] / dT[k][i]);
                if (dL[k][i] < 0) a = std::min(a, -lam[k][i] / dL[k][i]);
                if (any_soft && is_soft(k, (int)i)) {
                    if (dTau[k][i] < 0) a = std::min(a, -tau[k][i] / dTau[k][i]);
                    if (dGam[k][i] < 0) a = std::min(a, -gam[k][i] / dGam[k][i]);
                }
            }
            return a;
        };
        // predictor
        for (int k = 0; k <= N; ++k) for (size_t i = 0; i < t[k].size(); ++i) { rc[k][i] = lam[k][i] * t[k][i]; rcs[k][i] = gam[k][i] * tau[k][i]; }
        newton(dt_aff, dlam_aff, dtau_aff, dgam_aff);
        double a_aff = max_step(dt_aff, dlam_aff, dtau_aff, dgam_aff);
        double mu_aff = 0;
        for (int k = 0; k <= N; ++k) for (size_t i = 0; i < t[k].size(); ++i) {
            mu_aff += (lam[k][i] + a_aff * dlam_aff[k][i]) * (t[k][i] + a_aff * dt_aff[k][i]);
            if (any_soft && is_soft(k, (int)i)) mu_aff += (gam[k][i] + a_aff * dgam_aff[k][i]) * (tau[k][i] + a_aff * dtau_aff[k][i]);
        }
        mu_aff /= std::max(1, ntot);
        double sigma = std::pow(mu_aff / mu, 3.0);
        // keep the complementarity target from collapsing far below the tolerance while the other
        // residuals are still converging (weights lam/t would overflow the factorisation otherwise)
        if (sigma * mu < 1e-2 * tol) sigma = 1e-2 * tol / mu;
        // corrector
        for (int k = 0; k <= N; ++k) for (size_t i = 0; i < t[k].size(); ++i) {
            rc[k][i] = lam[k][i] * t[k][i] + dt_aff[k][i] * dlam_aff[k][i] - sigma * mu;
            rcs[k][i] = gam[k][i] * tau[k][i] + dtau_aff[k][i] * dgam_aff[k][i] - sigma * mu;
        }
        newton(dt_, dlam, dtau, dgam);
        double alpha = std::min(1.0, 0.995 * max_step(dt_, dlam, dtau, dgam));
        // Centrality safeguard of the step length in the first iterations (the wide neighbourhood N_-inf(gamma) of infeasible
        // path-following methods: S. Wright, Primal-Dual Interior-Point Methods, ch. 6; Mehrotra 1992, sec. 6): while the iterate
        // is still far from the central path (it < 4: the infeasible start) the step is shortened once, by 0.9, if a
        // complementarity product of the trial point lies below gamma = 0.02 times their average.  Without it a few rows run
        // ahead to the boundary in the first iterations and the later steps stall at alpha ~ 0.4 (two of the 1024 headline
        // instances needed 16 iterations and three 14; now none more than 12, the mean is unchanged).  Same constants in the
        // kernels (UPR_QP_NGAM, UPR_QP_NBT, UPR_QP_NIT in upr_qp.h).
        {
#ifndef ORC_NGAM_DEFAULT
#define ORC_NGAM_DEFAULT 0.02   /* oracle/Makefile builds a second library with 0.0: the same QPs along a DIFFERENT interior-point path (tests: the minimiser does not depend on the path) */
#endif
            static const double ngam = getenv("ORC_NGAM") ? atof(getenv("ORC_NGAM")) : ORC_NGAM_DEFAULT;
            static const int nit = getenv("ORC_NIT") ? atoi(getenv("ORC_NIT")) : 4;
            if (ngam > 0.0 && it < nit) {
                double mn = 1e300, sm = 0.0;
                for (int k = 0; k <= N; ++k) for (size_t i = 0; i < t[k].size(); ++i) {
                    const double v = (lam[k][i] + alpha * dlam[k][i]) * (t[k][i] + alpha * dt_[k][i]);
                    mn = std::min(mn, v); sm += v;
                    if (any_soft && is_soft(k, (int)i)) { const double vs = (gam[k][i] + alpha * dgam[k][i]) * (tau[k][i] + alpha * dtau[k][i]); mn = std::min(mn, vs); sm += vs; }
                }
                if (!(mn >= ngam * (sm / std::max(1, ntot)))) alpha *= 0.9;
            }
        }
        for (int k = 0; k <= N; ++k) {
            if (k >= 1) for (int i = 0; i < nx; ++i) dx[k][i] += alpha * ddx[k][i];
            if (k < N) for (int i = 0; i < nu; ++i) du[k][i] += alpha * ddu[k][i];
            for (int i = 0; i < nx; ++i) pi[k][i] += alpha * (pi_new[k][i] - pi[k][i]);
            if (k < N) for (int r = 0; r < ne; ++r) nu_[k][r] += alpha * (nu_new[k][r] - nu_[k][r]);
            for (size_t i = 0; i < t[k].size(); ++i) { t[k][i] += alpha * dt_[k][i]; lam[k][i] += alpha * dlam[k][i]; }
            if (any_soft) for (size_t i = 0; i < t[k].size(); ++i) if (is_soft(k, (int)i)) { sig[k][i] += alpha * dsig[k][i]; tau[k][i] += alpha * dtau[k][i]; gam[k][i] += alpha * dgam[k][i]; }
        }
        for (int r = 0; r < neN; ++r) yN[r] += alpha * dyN[r];
    }
    sol.iters = it; sol.dx = dx; sol.du = du; sol.pi = pi; sol.nu_ = nu_; sol.yN = yN;
    // linear policy of the last Riccati factorisation: du = -(Kx + Y S^-1 Cbar) dx - (feed-forward)
    // ([UPSTREAM] hpipm_interface getRiccatiFeedback; consumed through ocs2::LinearController when
    // sqp.use_feedback_policy is set, upright_control/src/pybindings.cpp:199)
    sol.K.assign(N, vec());
    if (!ric.Kx.empty()) for (int k = 0; k < N; ++k) {
        if (ric.Kx[k].empty()) continue;
        vec Kt = ric.Kx[k];
        // (a factorisation that broke down part-way leaves some knots without all of their factors: no gain there)
        if (ne > 0 && !ric.Ls[k].empty() && !ric.Cbar[k].empty() && !ric.Y[k].empty()) {
            vec SC = ric.Cbar[k]; chol_solve(ric.Ls[k].data(), ne, SC.data(), nx);
            for (int i = 0; i < nu; ++i) for (int j = 0; j < nx; ++j) { double t = 0; for (int r = 0; r < ne; ++r) t += ric.Y[k][i * ne + r] * SC[r * nx + j]; Kt[i * nx + j] += t; }
        }
        for (double& v : Kt) v = -v;
        sol.K[k] = Kt;
    }
    return sol.status;
}

// ------------------------------------------------------------------------------------------------
// OCP -> QP at a trajectory (multiple shooting transcription, [UPSTREAM] ocs2_sqp semantics:
// intermediate cost and its derivatives scaled by dt, defects b = f(x_k,u_k) - x_{k+1}).
void build_qp(const orc_problem* P, double t0, const double* x0, const double* xs, const double* us, QP& qp) {
    const int nx = orc_nx(P), nu = orc_nu(P), N = P->N, ne = 6 * P->nb, np = (P->nf == 3 ? 5 * P->nc : 0);
    qp.N = N; qp.nx = nx; qp.nu = nu; qp.ne = ne; qp.np = np; qp.nfc = nu - P->nq; qp.no = P->n_pairs + P->n_proj;
    qp.soft_x = P->soft_state_box != 0; qp.soft_u = P->soft_input_box != 0; qp.soft_poly = P->soft_poly != 0; qp.soft_eq = P->soft_eq != 0;
    qp.ZL = P->soft_L2_lower; qp.ZU = P->soft_L2_upper; qp.zL = P->soft_L1_lower; qp.zU = P->soft_L1_upper;
    Dyn dyn{P->nq, nx, nu, P->dt};
    dyn.dense(qp.A, qp.B);
    qp.st.assign(N, StageQP());
    qp.dx0.assign(nx, 0.0);
    for (int i = 0; i < nx; ++i) qp.dx0[i] = x0[i] - xs[i];
    vec gxm((size_t)ne * nx), gum((size_t)ne * nu), g(ne), grad_x(nx), grad_u(nu), Hxx((size_t)nx * nx), Huu(nu), h(np > 0 ? np : 1);
    // constant friction-row Jacobian: rows of E = d h / d u
    vec Gu((size_t)np * nu, 0.0);
    if (np > 0) {
        vec uz(nu, 0.0), h0(np), h1(np);
        orc_ineq_constraint(P, uz.data(), h0.data());
        for (int j = P->nq; j < nu; ++j) { uz[j] = 1.0; orc_ineq_constraint(P, uz.data(), h1.data()); uz[j] = 0.0; for (int r = 0; r < np; ++r) Gu[r * nu + j] = h1[r] - h0[r]; }
    }
    for (int k = 0; k < N; ++k) {
        StageQP& s = qp.st[k];
        const double* x = xs + (size_t)k * nx; const double* u = us + (size_t)k * nu;
        double t = t0 + k * P->dt;
        orc_stage_cost(P, t, x, u, grad_x.data(), grad_u.data(), Hxx.data(), Huu.data());
        s.Q.resize((size_t)nx * nx); s.q.resize(nx); s.Rd.resize(nu); s.r.resize(nu);
        for (size_t i = 0; i < s.Q.size(); ++i) s.Q[i] = P->dt * Hxx[i];
        for (int i = 0; i < nx; ++i) s.q[i] = P->dt * grad_x[i];
        for (int i = 0; i < nu; ++i) { s.Rd[i] = P->dt * Huu[i]; s.r[i] = P->dt * grad_u[i]; }
        s.b.assign(nx, 0.0);
        dyn.Ax(x, s.b.data()); dyn.Bu_add(u, s.b.data());
        for (int i = 0; i < nx; ++i) s.b[i] -= xs[(size_t)(k + 1) * nx + i];
        if (ne > 0) { orc_eq_constraint(P, x, u, g.data(), gxm.data(), gum.data()); s.Ce = gxm; s.De = gum; s.e = g; }
        s.xl.resize(nx); s.xu.resize(nx); s.ul.resize(nu); s.uu.resize(nu);
        for (int i = 0; i < nx; ++i) { s.xl[i] = P->x_lb[i] - x[i]; s.xu[i] = P->x_ub[i] - x[i]; }
        for (int i = 0; i < nu; ++i) { s.ul[i] = P->u_lb[i] - u[i]; s.uu[i] = P->u_ub[i] - u[i]; }
        if (np > 0) { orc_ineq_constraint(P, u, h.data()); s.Gu = Gu; s.d.assign(h.begin(), h.begin() + np); }
        if (qp.no > 0 && k >= 1) {   // collision rows: d(q) + dd/dq dq >= 0
            s.gd.assign(qp.no, 0.0); s.Gx.assign((size_t)qp.no * nx, 0.0);
            vec dq((size_t)qp.no * P->nq);
            orc_obstacle_rows(P, x, k * P->dt, s.gd.data(), dq.data());
            for (int r = 0; r < qp.no; ++r) for (int j = 0; j < P->nq; ++j) s.Gx[r * nx + j] = dq[r * P->nq + j];
        }
    }
    const double* xN = xs + (size_t)N * nx;
    qp.QN.assign((size_t)nx * nx, 0.0); qp.qN.assign(nx, 0.0);  // no final cost (controller_interface.cpp:138-148)
    qp.xlN.resize(nx); qp.xuN.resize(nx);
    for (int i = 0; i < nx; ++i) { qp.xlN[i] = P->x_lb[i] - xN[i]; qp.xuN[i] = P->x_ub[i] - xN[i]; }
    if (P->terminal_constraint) {
        qp.neN = 3 + 2 * P->nq; qp.CN.resize((size_t)qp.neN * nx); qp.eN.resize(qp.neN);
        orc_terminal_constraint(P, t0 + N * P->dt, xN, qp.eN.data(), qp.CN.data());
    } else { qp.neN = 0; qp.CN.clear(); qp.eN.clear(); }
}

}  // namespace

// ================================================================================================
extern "C" {

int orc_nx(const orc_problem* P) { return 3 * P->nq; }
int orc_nu(const orc_problem* P) { return P->nq + P->nf * P->nc; }

void orc_sphere_centers(const orc_problem* P, const double* x, double* c) {
    std::vector<V3<double>> cs;
    sphere_centers<double>(P, x, 0.0, cs);
    for (int s = 0; s < P->n_sph; ++s) for (int i = 0; i < 3; ++i) c[3 * s + i] = cs[s][i];
}

void orc_obstacle_rows(const orc_problem* P, const double* x, double tau, double* d, double* dq) {
    const int nq = P->nq, np = P->n_pairs + P->n_proj;
    if (np == 0) return;
    obstacle_rows<double>(P, x, tau, d);
    if (!dq) return;
    std::vector<Dual> xd(nq), dd(np);
    for (int j = 0; j < nq; ++j) {
        for (int i = 0; i < nq; ++i) xd[i] = Dual(x[i], i == j ? 1.0 : 0.0);
        obstacle_rows<Dual>(P, xd.data(), tau, dd.data());
        for (int r = 0; r < np; ++r) dq[r * nq + j] = dd[r].d;
    }
}

void orc_object_dynamics(const orc_problem* P, const double* forces, const double* C, const double* w,
                         const double* al, const double* a, double* out) {
    EEState<double> X;
    X.C = from_d<double>(C); X.w = from_d3<double>(w); X.al = from_d3<double>(al); X.a = from_d3<double>(a);
    X.p = {{0, 0, 0}}; X.v = {{0, 0, 0}};
    object_dynamics<double>(P, forces, X, out);
}

void orc_friction_rows(const orc_problem* P, const double* forces, double* out) { friction_rows<double>(P, forces, out); }

static void pack_state(const EEState<double>& S, double* o) {
    for (int i = 0; i < 3; ++i) { o[i] = S.p[i]; o[12 + i] = S.v[i]; o[15 + i] = S.w[i]; o[18 + i] = S.a[i]; o[21 + i] = S.al[i]; }
    for (int i = 0; i < 9; ++i) o[3 + i] = S.C.m[i];
}

void orc_ee_kinematics(const orc_problem* P, const double* x, double* out, double* dout) {
    const int nx = orc_nx(P);
    EEState<double> S = ee_kin<double>(P, x);
    pack_state(S, out);
    if (!dout) return;
    std::vector<Dual> xd(nx);
    for (int j = 0; j < nx; ++j) {
        for (int i = 0; i < nx; ++i) xd[i] = Dual(x[i], i == j ? 1.0 : 0.0);
        EEState<Dual> D = ee_kin<Dual>(P, xd.data());
        double col[24];
        for (int i = 0; i < 3; ++i) { col[i] = D.p[i].d; col[12 + i] = D.v[i].d; col[15 + i] = D.w[i].d; col[18 + i] = D.a[i].d; col[21 + i] = D.al[i].d; }
        for (int i = 0; i < 9; ++i) col[3 + i] = D.C.m[i].d;
        for (int r = 0; r < 24; ++r) dout[r * nx + j] = col[r];
    }
}

void orc_eq_constraint(const orc_problem* P, const double* x, const double* u, double* g, double* gx, double* gu) {
    const int nx = orc_nx(P), nu = orc_nu(P), ne = 6 * P->nb;
    eq_con<double>(P, x, u, g);
    if (!gx && !gu) return;
    std::vector<Dual> xd(nx), ud(nu), gd(ne);
    for (int i = 0; i < nx; ++i) xd[i] = Dual(x[i]);
    for (int i = 0; i < nu; ++i) ud[i] = Dual(u[i]);
    if (gx) for (int j = 0; j < nx; ++j) {
        xd[j].d = 1.0; eq_con<Dual>(P, xd.data(), ud.data(), gd.data()); xd[j].d = 0.0;
        for (int r = 0; r < ne; ++r) gx[r * nx + j] = gd[r].d;
    }
    if (gu) for (int j = 0; j < nu; ++j) {
        ud[j].d = 1.0; eq_con<Dual>(P, xd.data(), ud.data(), gd.data()); ud[j].d = 0.0;
        for (int r = 0; r < ne; ++r) gu[r * nu + j] = gd[r].d;
    }
}

void orc_ineq_constraint(const orc_problem* P, const double* u, double* h) {
    if (P->nf != 3) return;  // controller_interface.cpp:333-349: friction rows only with 3-D forces
    friction_rows<double>(P, u + P->nq, h);
}

double orc_stage_cost(const orc_problem* P, double t, const double* x, const double* u, double* grad_x,
                      double* grad_u, double* Hxx, double* Huu_diag) {
    const int nx = orc_nx(P), nu = orc_nu(P);
    double c = 0;
    for (int i = 0; i < nx; ++i) { double e = x[i] - P->xd[i]; c += 0.5 * P->Qdiag[i] * e * e; }
    for (int i = 0; i < nu; ++i) c += 0.5 * P->Rdiag[i] * u[i] * u[i];
    double pd[3]; target_position(P, t, pd);
    EEState<double> S = ee_kin<double>(P, x);
    const bool ori = P->Wee[3] != 0 || P->Wee[4] != 0 || P->Wee[5] != 0;
    const int nr = ori ? 6 : 3;
    double e[6] = {0, 0, 0, 0, 0, 0}, qd[4] = {0, 0, 0, 1};
    for (int r = 0; r < 3; ++r) e[r] = S.p[r] - pd[r];
    if (ori) {
        target_orientation(P, t, qd);
        const double nq_ = std::sqrt(qd[0] * qd[0] + qd[1] * qd[1] + qd[2] * qd[2] + qd[3] * qd[3]);
        for (int i = 0; i < 4; ++i) qd[i] /= nq_;
        orientation_error<double>(S.C, qd, e + 3);
    }
    for (int r = 0; r < nr; ++r) c += 0.5 * P->Wee[r] * e[r] * e[r];
    if (grad_x || Hxx) {
        std::vector<double> J((size_t)6 * nx, 0.0);
        std::vector<Dual> xd(nx);
        for (int i = 0; i < nx; ++i) xd[i] = Dual(x[i]);
        for (int j = 0; j < P->nq; ++j) {  // the pose depends on q only
            xd[j].d = 1.0; EEState<Dual> D = ee_kin<Dual>(P, xd.data()); xd[j].d = 0.0;
            for (int r = 0; r < 3; ++r) J[r * nx + j] = D.p[r].d;
            if (ori) { Dual eo[3]; orientation_error<Dual>(D.C, qd, eo); for (int r = 0; r < 3; ++r) J[(3 + r) * nx + j] = eo[r].d; }
        }
        if (grad_x) for (int i = 0; i < nx; ++i) {
            double g = P->Qdiag[i] * (x[i] - P->xd[i]);
            for (int r = 0; r < nr; ++r) g += J[r * nx + i] * P->Wee[r] * e[r];
            grad_x[i] = g;
        }
        if (Hxx) for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) {
            double h = (i == j ? P->Qdiag[i] : 0.0);
            for (int r = 0; r < nr; ++r) h += J[r * nx + i] * P->Wee[r] * J[r * nx + j];
            Hxx[i * nx + j] = h;
        }
    }
    if (grad_u) for (int i = 0; i < nu; ++i) grad_u[i] = P->Rdiag[i] * u[i];
    if (Huu_diag) for (int i = 0; i < nu; ++i) Huu_diag[i] = P->Rdiag[i];
    return c;
}

void orc_terminal_constraint(const orc_problem* P, double t, const double* x, double* c, double* cx) {
    const int nx = orc_nx(P), nq = P->nq, n = 3 + 2 * nq;
    double pd[3]; target_position(P, t, pd);
    EEState<double> S = ee_kin<double>(P, x);
    for (int r = 0; r < 3; ++r) c[r] = pd[r] - S.p[r];
    for (int i = 0; i < 2 * nq; ++i) c[3 + i] = x[nq + i];
    if (!cx) return;
    std::fill(cx, cx + (size_t)n * nx, 0.0);
    std::vector<Dual> xd(nx);
    for (int i = 0; i < nx; ++i) xd[i] = Dual(x[i]);
    for (int j = 0; j < nq; ++j) {
        xd[j].d = 1.0; EEState<Dual> D = ee_kin<Dual>(P, xd.data()); xd[j].d = 0.0;
        for (int r = 0; r < 3; ++r) cx[r * nx + j] = -D.p[r].d;
    }
    for (int i = 0; i < 2 * nq; ++i) cx[(3 + i) * nx + nq + i] = 1.0;
}

void orc_dynamics(const orc_problem* P, const double* x, const double* u, double* xnext) {
    Dyn dyn{P->nq, orc_nx(P), orc_nu(P), P->dt};
    dyn.Ax(x, xnext); dyn.Bu_add(u, xnext);
}

void orc_performance(const orc_problem* P, double t0, const double* x0, const double* xs, const double* us, double* out) {
    const int nx = orc_nx(P), nu = orc_nu(P), N = P->N, ne = 6 * P->nb, np = (P->nf == 3 ? 5 * P->nc : 0);
    double cost = 0, dyn_sse = 0, eq_sse = 0, ineq_sse = 0;
    vec xn(nx), g(std::max(ne, 1)), h(std::max(np, 1));
    for (int i = 0; i < nx; ++i) { double d = x0[i] - xs[i]; dyn_sse += d * d; }  // initial-state defect
    for (int k = 0; k < N; ++k) {
        const double* x = xs + (size_t)k * nx; const double* u = us + (size_t)k * nu;
        cost += P->dt * orc_stage_cost(P, t0 + k * P->dt, x, u, nullptr, nullptr, nullptr, nullptr);
        orc_dynamics(P, x, u, xn.data());
        for (int i = 0; i < nx; ++i) { double d = xn[i] - xs[(size_t)(k + 1) * nx + i]; dyn_sse += P->dt * d * d; }
        if (ne > 0) { orc_eq_constraint(P, x, u, g.data(), nullptr, nullptr); for (int r = 0; r < ne; ++r) eq_sse += P->dt * g[r] * g[r]; }
        if (np > 0) { orc_ineq_constraint(P, u, h.data()); for (int r = 0; r < np; ++r) { double v = std::min(0.0, h[r]); ineq_sse += P->dt * v * v; } }
        for (int i = 0; i < nu; ++i) { double v = std::min(0.0, std::min(u[i] - P->u_lb[i], P->u_ub[i] - u[i])); ineq_sse += P->dt * v * v; }
        if (k >= 1) for (int i = 0; i < nx; ++i) { double v = std::min(0.0, std::min(x[i] - P->x_lb[i], P->x_ub[i] - x[i])); ineq_sse += P->dt * v * v; }
        if (k >= 1 && P->n_pairs + P->n_proj > 0) { vec dd(P->n_pairs + P->n_proj); orc_obstacle_rows(P, x, k * P->dt, dd.data(), nullptr); for (double v : dd) { v = std::min(0.0, v); ineq_sse += P->dt * v * v; } }
    }
    const double* xN = xs + (size_t)N * nx;
    for (int i = 0; i < nx; ++i) { double v = std::min(0.0, std::min(xN[i] - P->x_lb[i], P->x_ub[i] - xN[i])); ineq_sse += v * v; }
    if (P->terminal_constraint) {
        vec c(3 + 2 * P->nq);
        orc_terminal_constraint(P, t0 + N * P->dt, xN, c.data(), nullptr);
        for (double v : c) eq_sse += v * v;
    }
    out[0] = cost; out[1] = dyn_sse; out[2] = eq_sse; out[3] = ineq_sse;
}

int orc_qp_step(const orc_problem* P, double t0, const double* x0, const double* xs, const double* us,
                double* dxs, double* dus, orc_stats* stats) {
    QP qp; build_qp(P, t0, x0, xs, us, qp);
    QPSol sol; int st = ipm_solve(qp, P->qp_iter_max, P->qp_tol, P->qp_tol_stat, sol);
    const int nx = qp.nx, nu = qp.nu;
    for (int k = 0; k <= qp.N; ++k) for (int i = 0; i < nx; ++i) dxs[(size_t)k * nx + i] = sol.dx[k][i];
    for (int k = 0; k < qp.N; ++k) for (int i = 0; i < nu; ++i) dus[(size_t)k * nu + i] = sol.du[k][i];
    if (stats) { stats->qp_iters_last = sol.iters; stats->qp_status_last = st; for (int i = 0; i < 4; ++i) stats->qp_res[i] = sol.res[i]; }
    return st;
}

// QP step + feedback gains K[N][nu][nx] of its last factorisation
int orc_qp_feedback(const orc_problem* P, double t0, const double* x0, const double* xs, const double* us,
                    double* dxs, double* dus, double* K, orc_stats* stats) {
    QP qp; build_qp(P, t0, x0, xs, us, qp);
    QPSol sol; int st = ipm_solve(qp, P->qp_iter_max, P->qp_tol, P->qp_tol_stat, sol);
    const int nx = qp.nx, nu = qp.nu;
    for (int k = 0; k <= qp.N; ++k) for (int i = 0; i < nx; ++i) dxs[(size_t)k * nx + i] = sol.dx[k][i];
    for (int k = 0; k < qp.N; ++k) for (int i = 0; i < nu; ++i) dus[(size_t)k * nu + i] = sol.du[k][i];
    for (int k = 0; k < qp.N; ++k) for (int e = 0; e < nu * nx; ++e) K[(size_t)k * nu * nx + e] = sol.K[k].empty() ? 0.0 : sol.K[k][e];
    if (stats) { stats->qp_iters_last = sol.iters; stats->qp_status_last = st; for (int i = 0; i < 4; ++i) stats->qp_res[i] = sol.res[i]; }
    return st;
}

// [UPSTREAM] ocs2_sqp MultipleShootingSolver::runImpl / takeStep / checkConvergence semantics:
// filter line search with alpha_decay 0.5, alpha_min 1e-4, gamma_c 1e-6, g_max 1e6, g_min 1e-6,
// armijo factor 1e-4 (ocs2_sqp defaults; none of them is bound at pybindings.cpp:190-213).
int orc_solve(const orc_problem* P, double t0, const double* x0, double* xs, double* us, orc_stats* stats) {
    const int nx = orc_nx(P), nu = orc_nu(P), N = P->N;
    const double alpha_decay = 0.5, alpha_min = 1e-4, gamma_c = 1e-6, g_max = 1e6, g_min = 1e-6, armijo = 1e-4;
    vec dxs((size_t)(N + 1) * nx), dus((size_t)N * nu), xt((size_t)(N + 1) * nx), ut((size_t)N * nu);
    orc_stats st; std::memset(&st, 0, sizeof(st));
    for (int i = 0; i < nx; ++i) xs[i] = x0[i];  // multiple shooting: first node is the observation
    int iter = 0;
    for (; iter < P->sqp_iters; ++iter) {
        double base[4]; orc_performance(P, t0, x0, xs, us, base);
        double base_viol = std::sqrt(base[1] + base[2] + base[3]);
        QP qp; build_qp(P, t0, x0, xs, us, qp);
        QPSol sol; st.qp_status_last = ipm_solve(qp, P->qp_iter_max, P->qp_tol, P->qp_tol_stat, sol);
        st.qp_iters_last = sol.iters; for (int i = 0; i < 4; ++i) st.qp_res[i] = sol.res[i];
        if (st.qp_status_last == 2) break;
        double descent = 0, dxn = 0, dun = 0;  // armijo descent metric: cost gradient . step
        for (int k = 0; k <= N; ++k) for (int i = 0; i < nx; ++i) { double d = sol.dx[k][i]; dxs[(size_t)k * nx + i] = d; dxn += d * d; if (k < N) descent += qp.st[k].q[i] * d; }
        for (int k = 0; k < N; ++k) for (int i = 0; i < nu; ++i) { double d = sol.du[k][i]; dus[(size_t)k * nu + i] = d; dun += d * d; descent += qp.st[k].r[i] * d; }
        dxn = std::sqrt(dxn); dun = std::sqrt(dun);
        double alpha = 1.0; bool accepted = false; double perf[4] = {0, 0, 0, 0};
        do {
            for (size_t i = 0; i < xt.size(); ++i) xt[i] = xs[i] + alpha * dxs[i];
            for (size_t i = 0; i < ut.size(); ++i) ut[i] = us[i] + alpha * dus[i];
            orc_performance(P, t0, x0, xt.data(), ut.data(), perf);
            double viol = std::sqrt(perf[1] + perf[2] + perf[3]);
            if (viol > g_max) accepted = false;
            else if (viol < g_min) {
                if (descent < 0) accepted = perf[0] < base[0] + armijo * alpha * descent;
                else accepted = true;
            } else accepted = (perf[0] < base[0] - gamma_c * base_viol) || (viol < (1.0 - gamma_c) * base_viol);
            if (accepted) break;
            alpha *= alpha_decay;
        } while (alpha >= alpha_min);
        st.step_alpha_last = accepted ? alpha : 0.0;
        st.dx_norm = dxn; st.du_norm = dun;
        if (accepted) {
            std::copy(xt.begin(), xt.end(), xs); std::copy(ut.begin(), ut.end(), us);
            st.cost = perf[0]; st.constraint_violation = std::sqrt(perf[1] + perf[2] + perf[3]);
        } else { st.cost = base[0]; st.constraint_violation = base_viol; }
        st.sqp_iters_done = iter + 1;
        if (!accepted) break;                                                                   // STEPSIZE
        if (std::fabs(base[0] - st.cost) < P->cost_tol && st.constraint_violation < g_min) break; // METRICS
        if (alpha * dxn < P->delta_tol && alpha * dun < P->delta_tol) break;                     // PRIMAL
    }
    if (stats) *stats = st;
    return st.qp_status_last;
}

// n independent solves of one problem family, OpenMP over instances (the all-core CPU baseline --
// what the reference's sequential sweep, planning_sim_loop.py:613-655, becomes with one controller per core).
// Per instance: way_p[n][n_way][3] targets and body_params[n][nb][10] (either may be NULL = the family's), t0[n],
// x0[n][nx]; xs[n][(N+1) nx] / us[n][N nu] hold the guesses on entry and the solutions on exit.  Every thread works on
// its own copy of the problem; nthreads <= 0 = all the OpenMP runtime offers.  Returns the number of threads used.
int orc_solve_batch(const orc_problem* P, int n, const double* way_p, const double* body_params, const double* t0,
                    const double* x0, double* xs, double* us, orc_stats* stats, int nthreads) {
    const int nx = orc_nx(P), nu = orc_nu(P), N = P->N;
    int used = 1;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    used = nthreads;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        orc_problem Q = *P;   // thread-private family
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int b = 0; b < n; ++b) {
            if (way_p) std::memcpy(Q.way_p, way_p + (size_t)b * P->n_way * 3, sizeof(double) * P->n_way * 3);
            if (body_params) std::memcpy(Q.body_params, body_params + (size_t)b * P->nb * 10, sizeof(double) * P->nb * 10);
            orc_solve(&Q, t0[b], x0 + (size_t)b * nx, xs + (size_t)b * (N + 1) * nx, us + (size_t)b * N * nu, stats ? stats + b : nullptr);
        }
    }
    return used;
}

}  // extern "C"
