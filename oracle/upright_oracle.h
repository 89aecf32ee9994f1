/* upright_oracle.h -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (upright_amd/, libupright_mi.so) never links, imports or calls it.
 *
 * PARITY STATUS
 *   - setup-time data (bodies, contacts) is PINNED: tests/golden/arrangements.json is produced by
 *     importing the reference's own Python (tests/golden/make_fixtures.py) and matches the
 *     known answers in upright_core/tests/test_parsing.py.
 *   - orc_object_dynamics / orc_friction_rows restate upright_core/include/upright_core/
 *     contact_constraints.h:50-194 line by line; the headers need Eigen (absent here), so they
 *     cannot be compiled into oracle/_ref: checked by physics known answers + a numpy twin.
 *   - kinematics, SQP and QP live in un-vendored, un-pinned third-party code (utiasDSL/ocs2
 *     branch `upright`, Pinocchio, HPIPM; /root/reference/README.md:44-51):  "parity unpinned".
 *     The restatement follows the published algorithms (multiple-shooting SQP with filter line
 *     search, Mehrotra predictor-corrector IPM over a stage-wise Riccati factorisation) and the
 *     reference's call sites (upright_control/src/controller_interface.cpp:103-393).
 */
#ifndef UPRIGHT_ORACLE_H
#define UPRIGHT_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_JOINTS 16
#define ORC_MAX_CONTACTS 64
#define ORC_MAX_BODIES 16
#define ORC_MAX_WAYPOINTS 8
#define ORC_MAX_NX 64
#define ORC_MAX_NU 208
#define ORC_MAX_SPHERES 32
#define ORC_MAX_PAIRS 64
#define ORC_MAX_DYN 4

typedef struct {
    /* ---- dimensions (upright_control/include/upright_control/dimensions.h:18-46) ---- */
    int nq;  /* robot joints: nq == nv; state x = [q, v, a] (3*nq) */
    int nb;  /* balanced bodies */
    int nc;  /* contact points */
    int nf;  /* force variables per contact: 3 (friction) or 1 (frictionless) */
    int N;   /* shooting intervals; N+1 knots */
    double dt;

    /* ---- kinematic chain: joint i frame = parent frame * (R_i, p_i) * motion(axis_i, q_i) ---- */
    int joint_type[ORC_MAX_JOINTS]; /* 0 = prismatic, 1 = revolute */
    double joint_axis[ORC_MAX_JOINTS][3];
    double joint_R[ORC_MAX_JOINTS][9]; /* row-major */
    double joint_p[ORC_MAX_JOINTS][3];
    double tool_R[9];
    double tool_p[3];

    double gravity[3];

    /* ---- bodies: 10 inertial parameters each [m, m*c, Ixx,Ixy,Ixz,Iyy,Iyz,Izz]
     *      (upright_core/include/upright_core/rigid_body.h:36-51), std::map (sorted-name) order */
    double body_params[ORC_MAX_BODIES][10];

    /* ---- contacts (upright_core/include/upright_core/contact.h:10-46) ---- */
    int contact_body1[ORC_MAX_CONTACTS]; /* index into bodies, -1 = EE / fixture (not balanced) */
    int contact_body2[ORC_MAX_CONTACTS];
    double contact_mu[ORC_MAX_CONTACTS];
    double contact_normal[ORC_MAX_CONTACTS][3];
    double contact_span[ORC_MAX_CONTACTS][6]; /* 2x3 row-major */
    double contact_r1[ORC_MAX_CONTACTS][3];
    double contact_r2[ORC_MAX_CONTACTS][3];

    /* ---- cost (controller_interface.cpp:400-420, cost/end_effector_cost.h:33-84) ---- */
    double Qdiag[ORC_MAX_NX];  /* state weight diagonal */
    double Rdiag[ORC_MAX_NU];  /* input weight diagonal (jerk weights then force_weight) */
    double xd[ORC_MAX_NX];     /* desired joint state */
    double Wee[6];             /* EE pose weight diagonal: position (3), orientation error (3) */

    /* ---- bounds (controller_interface.cpp:157-169,330-357) ---- */
    double x_lb[ORC_MAX_NX], x_ub[ORC_MAX_NX];
    double u_lb[ORC_MAX_NU], u_ub[ORC_MAX_NU];

    /* ---- target (wrappers.py:26-43; reference_trajectory.h:18-47) ---- */
    int n_way;
    double way_t[ORC_MAX_WAYPOINTS];
    double way_p[ORC_MAX_WAYPOINTS][3];
    double way_q[ORC_MAX_WAYPOINTS][4]; /* target orientations, xyzw (reference_trajectory.h:14-16); used when Wee[3..5] != 0 */

    /* ---- solver settings (controller.yaml:54-72) ---- */
    int sqp_iters;
    int qp_iter_max;
    double qp_tol;       /* residual tolerance of the IPM */
    double delta_tol;    /* controller.yaml:58 */
    double cost_tol;     /* controller.yaml:59 */
    int terminal_constraint; /* 1: stationary_desired_position_constraint at knot N */

    /* ---- collision avoidance (controller_interface.cpp:172-228,450-481): spheres rigidly attached to chain frames
     *      (frame = -1 world, i < nq the link after joint i, nq the tool frame) and the pairs whose distance
     *      |c_a - c_b| - r_a - r_b - obs_min_dist must stay >= 0 at knots 1..N-1 ([UPSTREAM]
     *      ocs2::SelfCollisionConstraintCppAd over hpp-fcl sphere-sphere distances) ---- */
    int n_sph;
    int sph_frame[ORC_MAX_SPHERES];
    double sph_off[ORC_MAX_SPHERES][3];
    double sph_r[ORC_MAX_SPHERES];
    int n_pairs;
    int pair_a[ORC_MAX_PAIRS], pair_b[ORC_MAX_PAIRS];
    double obs_min_dist;
    /* pair_b == -1: the ground half-space z >= 0 (controller_interface.cpp:93-101): d = c_a.z - r_a - obs_min_dist.
     * sph_frame == -2 - i: dynamic obstacle i (system_dynamics.h:29-39: [r, v, a] with rdot = v, vdot = a, adot = 0; the
     * state carries 9 entries per obstacle, dimensions.h:32-45), whose OBSERVED state at the start of the horizon is
     * dyn_x0[9 i ..]; it is uncontrolled, so over the horizon it is the known ballistic function of time
     * r(tau) = r0 + tau v0 + tau^2/2 a0.  The projectile-path rows follow the LAST obstacle
     * (projectile_path_constraint.h:82,114: state.tail(9)). */
    int n_dyn;          /* 0 .. ORC_MAX_DYN */
    double dyn_x0[9 * ORC_MAX_DYN];
    /* projectile_path_constraint.h:12-167: rows w s (|c - r_closest| - dist_i), w = proj_scale / dist_i, c the centre of
     * sphere proj_sph[i], r_closest the closest FUTURE point of the obstacle's path (cubic by Newton, 10 steps, 1e-4) */
    int n_proj;
    int proj_sph[8];
    double proj_dist[8];
    double proj_scale;
    double proj_s;      /* activation flag of the target (8th entry of the target state) */
    /* soft constraints: ocs2 hpipm_interface SlackSettings (upright_control/src/pybindings.cpp:160-181; defaults of
     * upright_control/src/upright_control/wrappers.py:121-143: L2 100, L1 0, slack lower bound 0) */
    int soft_state_box, soft_input_box, soft_poly;
    double soft_L2_lower, soft_L2_upper, soft_L1_lower, soft_L1_upper;
    /* the object-dynamics equality softened too: ocs2's HPIPM interface enters state-input equalities as general
     * constraints with lg = ug, which `slacks.poly_ineq` covers.  Equal L2 penalties Z, no L1 penalty (the reference's
     * defaults): the slack pair of a row eliminates to the quadratic penalty Z/2 |C dx + D du + e|^2, solved here as
     * the regularised equality C dx + D du + e = nu / Z. */
    int soft_eq;
    double qp_tol_stat;  /* stationarity tolerance of the IPM (HPIPM tol_stat; ocs2 default 1e-6); <= 0: qp_tol */
} orc_problem;

int orc_nx(const orc_problem* P);
/* state rows at state x, tau seconds after the observation: d[n_pairs + n_proj] (collision pairs, then projectile
 * rows) and (if not NULL) dq[n_pairs + n_proj][nq] = d d / d q */
void orc_obstacle_rows(const orc_problem* P, const double* x, double tau, double* d, double* dq);
/* sphere centres c[n_sph][3] at configuration q = x[0:nq] */
void orc_sphere_centers(const orc_problem* P, const double* x, double* c);
int orc_nu(const orc_problem* P);

/* ---- upright_core.bindings twins (upright_core/src/pybindings.cpp:53-56), unnormalised ---- */
/* C row-major world<-EE, w/al angular vel/acc (world), a linear acc (world).  out[6*nb]. */
void orc_object_dynamics(const orc_problem* P, const double* forces, const double* C,
                         const double* w, const double* al, const double* a, double* out);
/* forces[3*nc] -> out[5*nc] */
void orc_friction_rows(const orc_problem* P, const double* forces, double* out);

/* ---- end-effector kinematics: out = [p(3), C(9 row-major), v(3), w(3), a(3), al(3)] = 24 values;
 *      dout (may be NULL) = d out / d x, 24 x nx row-major ---- */
void orc_ee_kinematics(const orc_problem* P, const double* x, double* out, double* dout);

/* ---- per-knot OCP terms at (t, x, u) ---- */
/* object_dynamics equality (balancing_constraints.cpp:114-155): g[6nb] (scaled by 1/sqrt(6nb)),
 * gx[6nb x nx], gu[6nb x nu] row-major (either Jacobian may be NULL) */
void orc_eq_constraint(const orc_problem* P, const double* x, const double* u, double* g,
                       double* gx, double* gu);
/* contact_forces inequality (balancing_constraints.cpp:59-71): h[5nc] >= 0 (nf == 3 only) */
void orc_ineq_constraint(const orc_problem* P, const double* u, double* h);
/* intermediate cost (not scaled by dt): state_input_cost + end_effector_cost at time t.
 * grad_x[nx], grad_u[nu], Hxx[nx*nx] (Gauss-Newton), Huu diag[nu] may be NULL */
double orc_stage_cost(const orc_problem* P, double t, const double* x, const double* u,
                      double* grad_x, double* grad_u, double* Hxx, double* Huu_diag);
/* terminal equality (stationary_desired_position_constraint.h:35-74): c[3+2nq], cx row-major */
void orc_terminal_constraint(const orc_problem* P, double t, const double* x, double* c, double* cx);
/* discrete dynamics x+ = A x + B u (system_dynamics.h:15-22 integrated exactly) */
void orc_dynamics(const orc_problem* P, const double* x, const double* u, double* xnext);

/* ---- solve ---- */
typedef struct {
    int sqp_iters_done;
    int qp_iters_last;
    int qp_status_last;   /* 0 converged, 1 max iter, 2 numerical failure */
    double step_alpha_last;
    double cost;          /* merit after last step */
    double constraint_violation; /* sqrt(dyn SSE + eq SSE + ineq SSE) after last step */
    double qp_res[4];     /* stationarity, equality, inequality, complementarity of last QP */
    double dx_norm, du_norm;
} orc_stats;

/* One MPC solve (MultipleShootingSolver::runImpl semantics): t0, x0 fixed; xs[(N+1)*nx],
 * us[N*nu] hold the initial guess on entry and the solution on exit. */
int orc_solve(const orc_problem* P, double t0, const double* x0, double* xs, double* us,
              orc_stats* stats);

/* n independent solves of one family with OpenMP over the instances (the all-core CPU baseline of bench.py).  way_p
 * [n][n_way][3] and body_params[n][nb][10] may be NULL (= the family's); nthreads <= 0: all.  Returns threads used. */
int orc_solve_batch(const orc_problem* P, int n, const double* way_p, const double* body_params, const double* t0,
                    const double* x0, double* xs, double* us, orc_stats* stats, int nthreads);

/* Build the QP of one SQP iteration at (xs, us) and solve it; dxs/dus get the step.
 * Exposed for QP-level parity tests. */
int orc_qp_step(const orc_problem* P, double t0, const double* x0, const double* xs,
                const double* us, double* dxs, double* dus, orc_stats* stats);
/* same + the feedback gains K[N][nu][nx] (u = bias + K x) of the last Riccati factorisation */
int orc_qp_feedback(const orc_problem* P, double t0, const double* x0, const double* xs, const double* us,
                    double* dxs, double* dus, double* K, orc_stats* stats);

/* merit terms at a trajectory: out = [cost, dyn_sse, eq_sse, ineq_sse] */
void orc_performance(const orc_problem* P, double t0, const double* x0, const double* xs,
                     const double* us, double* out);

#ifdef __cplusplus
}
#endif
#endif
