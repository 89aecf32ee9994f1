/* upright_mi.h -- C-ABI of libupright_mi.so, the MI355X-native batched MPC engine.
 *
 * Drop-in boundary for the reference's hot path.  Every entry point names the
 * reference interface it replaces (paths relative to the reference repository root).  Plain C types
 * only: pointers, sizes, doubles.  All functions return 0 on success, non-zero on failure, and
 * upr_last_error() returns the message (the Python shim raises it as RuntimeError, mirroring how
 * the reference's std::runtime_error surfaces through pybind11).
 *
 * Device policy: every compute entry point runs hand-written HIP kernels on the current HIP device.
 * There is NO CPU fallback: without a GPU the calls fail with an error.
 */
#ifndef UPRIGHT_MI_H
#define UPRIGHT_MI_H

#ifdef __cplusplus
extern "C" {
#endif

#define UPR_MAX_JOINTS 12
#define UPR_MAX_CONTACTS 32
#define UPR_MAX_BODIES 8
#define UPR_MAX_WAYPOINTS 8
#define UPR_MAX_NX 36  /* 3 * UPR_MAX_JOINTS */
#define UPR_MAX_NU 108 /* UPR_MAX_JOINTS + 3 * UPR_MAX_CONTACTS */
#define UPR_MAX_SPHERES 16
#define UPR_MAX_PAIRS 32
#define UPR_MAX_DYN 4     /* dynamic obstacles (dimensions.h:32-45: the state carries 9 entries per obstacle) */

/* Problem family shared by all instances of a batch: what ControllerInterface's constructor reads
 * from ControllerSettings (upright_control/src/controller_interface.cpp:103-393;
 * upright_control/include/upright_control/controller_settings.h:47-119). */
typedef struct upr_problem {
    /* dims: upright_control/include/upright_control/dimensions.h:18-46 */
    int nq; /* robot joints (nq == nv); x = [q, v, a] */
    int nb; /* balanced bodies */
    int nc; /* contact points */
    int nf; /* 3 = frictional contact forces, 1 = frictionless (normal force only) */
    int N;  /* shooting intervals: time_horizon / dt (controller.yaml:13,55) */
    double dt;

    /* serial chain (replaces the Pinocchio model of upright_control/include/upright_control/util.h:15-65):
     * joint frame i = parent * (R_i, p_i) * motion(axis_i, q_i); type 0 prismatic, 1 revolute */
    int joint_type[UPR_MAX_JOINTS];
    double joint_axis[UPR_MAX_JOINTS][3];
    double joint_R[UPR_MAX_JOINTS][9];
    double joint_p[UPR_MAX_JOINTS][3];
    double tool_R[9];
    double tool_p[3];

    double gravity[3]; /* controller.yaml:6 */

    /* contacts: upright_core/include/upright_core/contact.h:10-46; body index -1 = EE / fixture */
    int contact_body1[UPR_MAX_CONTACTS];
    int contact_body2[UPR_MAX_CONTACTS];
    double contact_mu[UPR_MAX_CONTACTS];
    double contact_normal[UPR_MAX_CONTACTS][3];
    double contact_span[UPR_MAX_CONTACTS][6];
    double contact_r1[UPR_MAX_CONTACTS][3];
    double contact_r2[UPR_MAX_CONTACTS][3];

    /* costs: controller_interface.cpp:400-420, cost/end_effector_cost.h:33-84 */
    double Qdiag[UPR_MAX_NX];
    double Rdiag[UPR_MAX_NU];
    double xd[UPR_MAX_NX];
    double Wee[6]; /* diagonal of the end-effector pose weight: position (3), orientation error (3) */

    /* bounds: controller_interface.cpp:157-169,330-357 */
    double x_lb[UPR_MAX_NX], x_ub[UPR_MAX_NX];
    double u_lb[UPR_MAX_NU], u_ub[UPR_MAX_NU];

    /* target waypoint times (positions are per instance): reference_trajectory.h:18-47 */
    int n_way;
    double way_t[UPR_MAX_WAYPOINTS];

    /* solver settings: upright_control/src/pybindings.cpp:183-213, controller.yaml:54-72 */
    int sqp_iters;
    int qp_iter_max;
    double qp_tol;
    double delta_tol;
    double cost_tol;
    int terminal_constraint; /* stationary_desired_position_constraint at knot N */
    int use_feedback_policy; /* sqp.use_feedback_policy (controller.yaml:60; pybindings.cpp:199): keep the Riccati gains of
                                the last QP and apply them in upr_batch_evaluate_policy */

    /* collision avoidance (controller_interface.cpp:172-228,450-481; obstacles/simple.yaml:11-41): spheres rigidly
     * attached to chain frames (sph_frame: -1 world, i < nq the link carried by joint i, nq the tool frame) and the
     * pairs whose distance |c_a - c_b| - r_a - r_b - obs_min_dist stays >= 0 at knots 1..N-1 (hard state
     * inequality "obstacle_avoidance"; replaces ocs2::SelfCollisionConstraintCppAd over hpp-fcl sphere pairs) */
    int n_sph;
    int sph_frame[UPR_MAX_SPHERES];
    double sph_off[UPR_MAX_SPHERES][3];
    double sph_r[UPR_MAX_SPHERES];
    int n_pairs;
    int pair_a[UPR_MAX_PAIRS], pair_b[UPR_MAX_PAIRS];
    double obs_min_dist; /* controller.yaml:108 */
    /* pair_b == -1: the ground half-space z >= 0 (controller_interface.cpp:93-101).
     * n_dyn (0 .. UPR_MAX_DYN) dynamic obstacles (system_dynamics.h:29-39, dimensions.h:32-45; obstacles/dynamic.yaml): the
     * state handed to set_observation / returned by get_solution / evaluate is [robot x (3 nq), then r, v, a (9) of every
     * obstacle in turn]; an obstacle is uncontrolled, so inside the solve it is the ballistic function of time of its
     * observed state.  Spheres with sph_frame == -2 - i ride on obstacle i (-2: the first).  The projectile-path rows below
     * follow the LAST obstacle (projectile_path_constraint.h:82,114 read state.tail(9)). */
    int n_dyn;
    /* projectile_path_constraint.h:12-167 ("projectile_constraint"): rows w s (|c_i - r_closest| - dist_i),
     * w = proj_scale / dist_i, c_i the centre of sphere proj_sph[i], r_closest the closest future point of the
     * obstacle's path, s the per-instance activation flag (upr_batch_set_projectile_flag) */
    int n_proj;
    int proj_sph[8];
    double proj_dist[8];
    double proj_scale;
    /* soft constraints: ocs2 hpipm_interface SlackSettings (upright_control/src/pybindings.cpp:160-181; values
     * upright_control/src/upright_control/wrappers.py:121-143).  A softened row c(z) >= 0 becomes c(z) + sigma >= 0,
     * sigma >= 0, with cost 1/2 Z sigma^2 + z sigma (Z: L2, z: L1 penalty; "lower" for lower bounds and polytopic
     * rows, "upper" for upper bounds). */
    int soft_state_box, soft_input_box, soft_poly;
    double soft_L2_lower, soft_L2_upper, soft_L1_lower, soft_L1_upper;
    /* soft_eq: the object-dynamics equality is softened as well.  ocs2's HPIPM interface hands state-input equalities to
     * HPIPM as general (polytopic) constraints with lg = ug, so `slacks.poly_ineq` puts a slack pair on every such row:
     * lg - sl <= C dx + D du + e <= ug + su.  With equal L2 penalties Z and no L1 penalty (the reference's defaults,
     * wrappers.py:121-143) eliminating the pair leaves the quadratic penalty Z/2 |C dx + D du + e|^2, i.e. the
     * regularised equality C dx + D du + e = nu / Z -- the Schur complement S = Df Hff^-1 Df' + I / Z.  This is what
     * makes the frictionless multi-body problems of upright_robust (config/demos/_base.yaml:62-75) solvable at all:
     * with hard rows they admit no motion ("config 4").  Requires soft_L2_lower == soft_L2_upper > 0 and zero
     * L1 penalties. */
    int soft_eq;
    /* tolerance of the stationarity residual of the QP (HPIPM's tol_stat / res_g_max; ocs2's hpipm_interface::Settings
     * keeps it at 1e-6 next to 1e-8 for the equality, inequality and complementarity residuals, which `qp_tol` carries:
     * below ~1e-7 the stationarity residual of these problems is roundoff of the factorisation once the barrier
     * parameter is small).  <= 0: use qp_tol. */
    double qp_tol_stat;
} upr_problem;

const char* upr_last_error(void);
/* 1 if a HIP device is usable, else 0 (never initialises anything else) */
int upr_device_available(void);
/* One process per GPU (torch.distributed launchers hand every rank its LOCAL_RANK): select the HIP device the NEXT
 * upr_batch_create of this thread uses.  A handle remembers its device (upr_batch_device) and every call on it makes that device
 * current first.  The reference has no counterpart (one CPU solver per process). */
int upr_set_device(int device);

/* ------------------------------------------------------------------------------------------------
 * upright_core.bindings twins (upright_core/src/pybindings.cpp:53-56).  Batched: n states at once.
 * contact/body tables come from `P` (nq, chain and cost fields are ignored).
 *   upr_core_object_dynamics  <-> compute_object_dynamics_constraints (contact_constraints.h:162-194)
 *       forces[n][nf*nc], C[n][9] row-major world<-EE, w/al/a[n][3]  ->  out[n][6*nb] (UNnormalised)
 *   upr_core_friction_rows    <-> compute_contact_force_constraints_linearized (contact_constraints.h:50-77)
 *       forces[n][3*nc] -> out[n][5*nc]
 * body_params[nb][10] = [m, m*c, vech(I)] (rigid_body.h:47-51). Host pointers. */
int upr_core_object_dynamics(const upr_problem* P, const double* body_params, int n, const double* forces,
                             const double* C, const double* w, const double* al, const double* a, double* out);
int upr_core_friction_rows(const upr_problem* P, int n, const double* forces, double* out);

/* ------------------------------------------------------------------------------------------------
 * Batched MPC: B independent instances of the same problem family.  One instance with B = 1 is what
 * `bindings.ControllerInterface` (upright_control/src/pybindings.cpp:364-427) wraps.
 * Per-instance data: body_params[B][nb][10] (the upright_robust parameter axis,
 * balancing_constraints.cpp:92-102), way_p[B][n_way][3] (targets, wrappers.py:31-43). */
typedef struct upr_batch upr_batch;

upr_batch* upr_batch_create(const upr_problem* P, int B, const double* body_params, const double* way_p);
void upr_batch_destroy(upr_batch* h);

/* ControllerInterface.reset / setTargetTrajectories (pybindings.cpp:371-374): new targets, forget
 * the previous solution (next advance starts from the DefaultInitializer guess,
 * controller_interface.cpp:385-386). way_p may be NULL to keep targets. */
int upr_batch_reset(upr_batch* h, const double* way_p);

/* Target orientations (the quaternion part of ocs2::TargetTrajectories states, wrappers.py:31-43: Q_d = Q_EE(x0) (x) Q_offset):
 * way_q[B][n_way][4], xyzw; between waypoints the target is their SLERP (reference_trajectory.h:18-47).  They enter the
 * end-effector cost through its orientation error (cost/end_effector_cost.h:33-84) when Wee[3..5] != 0.  Default: identity. */
int upr_batch_set_target_orientations(upr_batch* h, const double* way_q);

/* ControllerInterface.setObservation (pybindings.cpp:369-370): t[B] (or t[1] broadcast if
 * t_stride == 0), x[B][nx_full], nx_full = 3 nq + 9 n_dyn. Host pointers.  Every x / xs argument of this interface has
 * nx_full entries per state; the feedback gains have nx_full columns (zero for the obstacle part). */
int upr_batch_set_observation(upr_batch* h, const double* t, int t_stride, const double* x);

/* Overwrite the initial guess (operating points, controller_interface.cpp:376-383): xs[B][N+1][nx],
 * us[B][N][nu]. */
int upr_batch_set_guess(upr_batch* h, const double* xs, const double* us);

/* ControllerInterface.advanceMpc (pybindings.cpp:375): one MPC solve = `sqp_iters` SQP iterations
 * (linearise -> QP -> filter line search) for every instance, all on the GPU. */
int upr_batch_advance(upr_batch* h);

/* SQP iterations of the NEXT advance only (afterwards `sqp_iters` of the problem again): ocs2's solver runs
 * sqp.init_sqp_iteration iterations while it has no previous solution -- the first solve after construction or reset, and
 * every solve with mpc.cold_start (upright_control/src/pybindings.cpp:148,194-195; controller.yaml:15,56-57;
 * upright_robust/config/demos/_base.yaml:64-65 sets 3).  n = 0 cancels. */
int upr_batch_set_sqp_iterations(upr_batch* h, int n);

/* Same, but inputs stay in HBM and no host synchronisation is done (used by bench.py so that the
 * timed region contains only device work); upr_batch_sync waits for completion. */
int upr_batch_advance_async(upr_batch* h);
int upr_batch_sync(upr_batch* h);

/* ControllerInterface.getMpcSolution (pybindings.cpp:376-377): ts[B][N+1], xs[B][N+1][nx], us[B][N][nu] */
int upr_batch_get_solution(upr_batch* h, double* ts, double* xs, double* us);

/* ControllerInterface.evaluateMpcSolution (pybindings.cpp:378-381): interpolate the stored solution
 * at time t[B] -> x_out[B][nx], u_out[B][nu] (feed-forward policy). */
int upr_batch_evaluate(upr_batch* h, const double* t, int t_stride, double* x_out, double* u_out);

/* Linear feedback policy of the last solve (sqp.use_feedback_policy = true: the primal solution carries an
 * ocs2::LinearController, evaluated by ControllerPythonInterface::evaluateMpcSolution with the CURRENT state,
 * controller_python_interface.h:46-55):
 *     u(t, x) = (1 - a) [u_j + K_j (x - x_j)] + a [u_j+1 + K_j+1 (x - x_j+1)],   t = t_j + a dt
 * K_j are the Riccati gains of the last QP (sign convention of ocs2: u = bias + K x).  x_obs[B][nx], outputs as
 * upr_batch_evaluate.  Needs use_feedback_policy != 0 at creation. */
int upr_batch_evaluate_policy(upr_batch* h, const double* t, int t_stride, const double* x_obs, double* x_out, double* u_out);
/* One control period of the reference's loop in ONE call and one synchronisation: what ControllerManager.step does with three
 * (manager.py:156-176: setObservation(t, x) -> advanceMpc() -> evaluateMpcSolution(t, x, x_opt, u_opt)), for callers that replan
 * every period.  The observation is uploaded once (the policy is evaluated at the state that was just observed, at its own time),
 * transfers go through the engine's pinned staging buffer, and the statistics of the solve come back with the result.
 * x[B][nx_full] observed states, x_out[B][nx_full], u_out[B][nu] as upr_batch_evaluate[_policy]; stats_out[B][UPR_NSTATS] or NULL.
 * Results are bit-identical to the three calls.
 * From the third period in a row that enqueues the same operations (warm start, same SQP iteration count, no event timing ...)
 * the period's stream operations -- copies in, prepare, linearise, QP, line search, policy, copies out -- are replayed as ONE HIP
 * graph launch (UPR_TICK_GRAPH=0: never); upr_batch_tick_graph_replays counts the periods served that way. */
int upr_batch_tick(upr_batch* h, const double* t, int t_stride, const double* x, double* x_out, double* u_out, double* stats_out);
long long upr_batch_tick_graph_replays(upr_batch* h);

/* ControllerInterface.getLinearFeedbackGain (pybindings.cpp:382-384) at the knots: K[B][N][nu][nx]; jerk rows from the
 * Riccati recursion, contact-force rows from the elimination of the object-dynamics equality
 * (f = f* - Hff^-1 Df' S^-1 C dx). */
int upr_batch_get_feedback(upr_batch* h, double* K);

/* activation flag of the projectile constraint per instance: the 8th entry of the target state
 * (reference_trajectory.h; set by the target-flag protocol of mrt_node.cpp:243-265).  s[B]; default 0. */
int upr_batch_set_projectile_flag(upr_batch* h, const double* s);

/* ControllerInterface.getLastSolveTime (pybindings.cpp:366), milliseconds of the last advance */
double upr_batch_last_solve_ms(const upr_batch* h);

/* per-instance statistics of the last advance: stats[B][UPR_NSTATS] =
 * [sqp_iters_done, qp_iters_last, qp_status_last, step_alpha_last, cost, constraint_violation,
 *  qp_res_stat, qp_res_eq, qp_res_ineq, qp_res_comp, dx_norm, du_norm] */
#define UPR_NSTATS 12
int upr_batch_get_stats(upr_batch* h, double* stats);
/* restore == 0: keep a copy of the statistics and of the QP dispatch keys of the last advance; restore != 0: put it back.  Brackets
 * a solver-level query that solves one more QP on the handle (upr_batch_qp_kkt behind ControllerInterface.valueFunction,
 * pybindings.cpp:398-403): afterwards the statistics describe the solve again (ocs2's getValueFunction does not disturb its solver). */
int upr_batch_hold_stats(upr_batch* h, int restore);

/* ------------------------------------------------------------------------------------------------
 * Term-level access (ControllerInterface.getStateInputEqualityConstraintValue("object_dynamics"),
 * getStateInputInequalityConstraintValue("contact_forces"), getCostValue, pybindings.cpp:414-424;
 * BalancingConstraintWrapper.getLinearApproximation, balancing_constraint_wrapper.h:45-60).
 * Evaluates the per-knot linearisation kernel on arbitrary (x, u) pairs of instance family `h`:
 *   n points, inst[n] instance index of each point (body parameters / target), t[n], x[n][nx], u[n][nu]
 *   g[n][ne], gx[n][ne][nx]             object_dynamics equality and d/dx   (ne = 6 nb)
 *   cost[n], grad[n][nq], hess[n][nq][nq]  end-effector cost, gradient and Gauss-Newton Hessian
 *   ee[n][3]                            end-effector position
 * Any output pointer may be NULL. */
int upr_batch_linearize_points(upr_batch* h, int n, const int* inst, const double* t, const double* x,
                               const double* u, double* g, double* gx, double* cost, double* grad,
                               double* hess, double* ee);
/* ControllerInterface.getStateInputInequalityConstraintValue("obstacle_avoidance", t, x, u) (pybindings.cpp:417-419;
 * mpc_sim.py:191-219) at n states: d[n][n_pairs] and (may be NULL) dq[n][n_pairs][nq] = d d / d q */
int upr_batch_obstacle_rows(upr_batch* h, int n, const double* x, double* d, double* dq);
/* constant d(object_dynamics)/du of instance `inst`: gu[ne][nu] */
int upr_batch_eq_input_jacobian(upr_batch* h, int inst, double* gu);

/* One QP (the SQP sub-problem at the current trajectory of every instance) without the line
 * search: dxs[B][N+1][nx], dus[B][N][nu].  For QP-level parity tests. */
int upr_batch_qp_step(upr_batch* h, double* dxs, double* dus);

/* The same QP with the multipliers it ended with, for an independent check of the optimality conditions
 * (tests/kkt_check.py assembles the QP in numpy from upr_batch_get_lin and the problem constants): pi[B][N+1][nx]
 * costates of the dynamics (pi_0 unused), nu[B][N][ne] multipliers of the object-dynamics rows, yN[B][3 + 2 nq] of the
 * terminal equality, lam[B][N+1][ni] of the inequality rows with ni = 2 nx + 2 nu + np + no in the slot order
 * [x lower][x upper][u lower][u upper][friction rows][collision / projectile rows]; *ni_out = ni.  Pointers may be NULL. */
int upr_batch_qp_kkt(upr_batch* h, double* dxs, double* dus, double* pi, double* nu, double* yN, double* lam, int* ni_out);
/* Slacks t[B][N+1][ni] of the inequality rows at the exit of the QP the last upr_batch_qp_kkt call solved (slot order of lam; 1 in
 * slots that are not rows of the knot).  lam / t are the barrier weights of the last interior-point iterate: with the costates they
 * are what the reference's solver-level queries -- valueFunction, valueFunctionStateDerivative,
 * stateInputEqualityConstraintLagrangian, upright_control/src/pybindings.cpp:398-412 (ocs2 getValueFunction = the Riccati
 * cost-to-go of the last QP, HPIPM's barrier-augmented P_k) -- are served from (upright_amd/value_function.py). */
int upr_batch_qp_slacks(upr_batch* h, double* t);

/* raw device pointers for zero-copy consumers (torch / RCCL all-gather of solved trajectories):
 * xs (B*(N+1)*nx doubles) and us (B*N*nu doubles) */
int upr_batch_device_ptrs(upr_batch* h, void** xs, void** us);

/* average device time (ms) of each kernel over the last advance, measured with HIP events on the
 * engine's stream: out[0] = linearise, out[1] = QP, out[2] = line search; launches[3] = #launches */
int upr_batch_kernel_times(upr_batch* h, double* ms, int* launches);
/* on: 0 no events; 1 events around every kernel; 2 around the QP kernel only; 3 around every fourth QP launch (the first, the
 * fifth, ...).  A recorded event holds the stream for 3 - 4 us (tools/exp_timing_modes.py; 6 us under rocprofv3, whose
 * kernel trace shows no gap between two kernels without one): a throughput run that wants the dominant kernel's average duration and
 * nothing else asks for 3 */
int upr_batch_enable_timing(upr_batch* h, int on);
/* name of the QP kernel instantiation this handle launches (as rocprofv3 prints it): bench.py's roofline.kernel */
const char* upr_batch_qp_kernel_name(const upr_batch* h);
int upr_batch_device(const upr_batch* h);   /* the HIP device the handle lives on; -1 for a null handle */
/* doubles of device workspace per instance (QP result, multipliers, the QP kernel's far arrays): what an instance writes once and
 * re-streams every interior-point iteration -- bench.py's model of the compulsory DRAM traffic of a launch */
long long upr_batch_ws_doubles(const upr_batch* h);

/* copy the current solution into caller-owned DEVICE buffers (torch tensors handed to the RCCL
 * all-gather of solved trajectories): xs_dst[B][N+1][nx], us_dst[B][N][nu]; asynchronous on the
 * engine's stream, follow with upr_batch_sync. */
int upr_batch_copy_solution_device(upr_batch* h, void* xs_dst, void* us_dst);
/* copy the inputs the last upr_batch_tick evaluated, u[B][nu] (the u_0 of every instance: what the closed loop's exchange
 * step gathers), from the engine's device buffer into a caller-owned DEVICE buffer; asynchronous on the engine's
 * stream.  Fails before the first tick. */
int upr_batch_copy_policy_device(upr_batch* h, void* u_dst);
/* the engine's HIP stream (a hipStream_t): for callers that order their own streams against it with events instead of
 * upr_batch_sync -- the RCCL exchange step of bench.py makes the collective's stream wait for the copy-out, no host sync */
void* upr_batch_stream(upr_batch* h);

/* debug / test accessors: per-phase cycle counters of the production QP kernel (first call arms it,
 * later calls read and clear prof[B][4][16]: 16 phases as seen by lane 0 of each of the first four waves); per-knot linearisation records lin[B][N+1][*stride] */
int upr_batch_qp_profile(upr_batch* h, double* out);
int upr_batch_get_lin(upr_batch* h, double* lin, int* stride);

/* upr_batch_reset without target change and without host synchronisation */
int upr_batch_reset_async(upr_batch* h);

#ifdef __cplusplus
}
#endif
#endif
