"""Host-side mirror of `upright_control/src/upright_control/{wrappers,manager}.py`:
same dict in, same settings out; `ControllerManager.step()` keeps the reference's replan cadence."""
import time

import numpy as np

from . import config as cfg
from . import control_bindings as bindings
from . import robots


def quat_multiply_xyzw(q0, q1):
    x0, y0, z0, w0 = q0
    x1, y1, z1, w1 = q1
    return np.array([
        w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1,
        w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1,
        w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1,
        w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1,
    ])


def rot_to_quat_xyzw(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    return q / np.linalg.norm(q)


class TargetTrajectories(bindings.TargetTrajectories):
    """wrappers.py:13-75."""

    @classmethod
    def from_config(cls, config, r_ew_w, Q_we, u):
        ts, xs, us = [], [], []
        for waypoint in config["waypoints"]:
            r_d = np.asarray(r_ew_w) + np.asarray(waypoint["position"], dtype=np.float64)
            Q_d = quat_multiply_xyzw(np.asarray(Q_we, dtype=np.float64), np.asarray(waypoint["orientation"], dtype=np.float64))
            ts.append(waypoint["time"])
            xs.append(np.concatenate((r_d, Q_d, [0])))
            us.append(np.copy(u))
        return cls(ts, xs, us)

    def poses(self):
        for x in self.xs:
            yield x[:3], x[3:7]

    def get_desired_pose(self, t):
        x = self.get_desired_state(t)
        return x[:3], x[3:7]


class ControllerSettings(bindings.ControllerSettings):
    """wrappers.py:78-399: controller config dict -> settings struct.  `bodies` / `contacts` default to what
    `upright_amd.arrangement.parse_control_objects` builds from the dict (the role of `parsing.py:351-410`); tests
    pass the reference's own parser output (tests/golden/arrangements.json) instead."""

    def __init__(self, config, x0=None, bodies=None, contacts=None):
        super().__init__()
        pn, pa = cfg.parse_number, cfg.parse_array
        self.mpc.time_horizon = pn(config["mpc"]["time_horizon"])
        self.mpc.debug_print = config["mpc"]["debug_print"]
        self.mpc.cold_start = config["mpc"]["cold_start"]
        r = config["rollout"]
        self.rollout.abs_tol_ode = pn(r["abs_tol_ode"]); self.rollout.rel_tol_ode = pn(r["rel_tol_ode"])
        self.rollout.timestep = pn(r["timestep"]); self.rollout.max_num_steps_per_second = pn(r["max_num_steps_per_second"], dtype=int)
        self.rollout.check_numerical_stability = r["check_numerical_stability"]
        q = config["sqp"]
        self.sqp.dt = pn(q["dt"]); self.sqp.sqp_iteration = q["sqp_iteration"]; self.sqp.init_sqp_iteration = q["init_sqp_iteration"]
        self.sqp.delta_tol = pn(q["delta_tol"]); self.sqp.cost_tol = pn(q["cost_tol"])
        self.sqp.use_feedback_policy = q["use_feedback_policy"]
        self.sqp.project_state_input_equality_constraints = q["project_state_input_equality_constraints"]
        self.sqp.print_solver_status = q["print_solver_status"]; self.sqp.print_solver_statistics = q["print_solver_statistics"]
        self.sqp.print_line_search = q["print_line_search"]
        hp = q["hpipm"]; sl = hp["slacks"]
        self.sqp.hpipm.warm_start = hp["warm_start"]; self.sqp.hpipm.iter_max = hp["iter_max"]
        s = self.sqp.hpipm.slacks
        s.enabled = sl["enabled"]; s.input_box = sl.get("input_box", True); s.state_box = sl.get("state_box", True); s.poly_ineq = sl.get("poly_ineq", True)
        s.upper_L2_penalty = sl.get("upper_L2_penalty", 100); s.lower_L2_penalty = sl.get("lower_L2_penalty", 100)
        s.upper_L1_penalty = sl.get("upper_L1_penalty", 0); s.lower_L1_penalty = sl.get("lower_L1_penalty", 0)
        s.upper_low_bound = sl.get("upper_low_bound", 0); s.lower_low_bound = sl.get("lower_low_bound", 0)
        self.end_effector_link_name = config["robot"]["tool_link_name"]
        self.robot_base_type = bindings.robot_base_type_from_string(config["robot"]["base_type"])
        e = config["estimation"]
        self.estimation.robot_init_variance = e["robot_init_variance"]; self.estimation.robot_process_variance = e["robot_process_variance"]
        self.estimation.robot_measurement_variance = e["robot_measurement_variance"]
        t = config["tracking"]
        for k in ("rate", "min_policy_update_time", "kp", "kv", "ka", "enforce_state_limits", "enforce_input_limits",
                  "enforce_ee_position_limits", "use_projectile", "state_violation_margin", "input_violation_margin",
                  "ee_position_violation_margin"):
            setattr(self.tracking, k, t[k])
        self.gravity = np.array(config["gravity"], dtype=np.float64)
        self.recompile_libraries = config.get("recompile_libraries", True)
        self.debug = config["debug"]
        d = config["robot"]["dims"]
        self.dims.robot.q, self.dims.robot.v, self.dims.robot.x, self.dims.robot.u = d["q"], d["v"], d["x"], d["u"]
        w = config["weights"]
        self.input_weight = cfg.parse_diag_matrix_dict(w["input"]); self.state_weight = cfg.parse_diag_matrix_dict(w["state"])
        self.end_effector_weight = cfg.parse_diag_matrix_dict(w["end_effector"])
        assert self.input_weight.shape == (self.dims.robot.u, self.dims.robot.u)
        assert self.state_weight.shape == (self.dims.robot.x, self.dims.robot.x)
        assert self.end_effector_weight.shape == (6, 6)
        lim = config["limits"]
        self.input_limit_lower = pa(lim["input"]["lower"]); self.input_limit_upper = pa(lim["input"]["upper"])
        self.state_limit_lower = pa(lim["state"]["lower"]); self.state_limit_upper = pa(lim["state"]["upper"])
        assert self.input_limit_lower.shape == (self.dims.robot.u,) and self.input_limit_upper.shape == (self.dims.robot.u,)
        assert self.state_limit_lower.shape == (self.dims.robot.x,) and self.state_limit_upper.shape == (self.dims.robot.x,)
        b = config["end_effector_box_constraint"]
        self.end_effector_box_constraint_enabled = b["enabled"]
        self.xyz_lower = pa(b["xyz_lower"]); self.xyz_upper = pa(b["xyz_upper"])
        if "projectile_path_constraint" in config:
            p = config["projectile_path_constraint"]
            self.projectile_path_constraint_enabled = p["enabled"]
            self.projectile_path_distances = np.array(p["distances"]); self.projectile_path_scale = p["scale"]
            self.projectile_path_collision_links = p["collision_links"]
        for name, value in config["robot"].get("locked_joints", {}).items():
            self.locked_joints[name] = pn(value)
        self.base_pose = np.array(config["robot"].get("base_pose", [0.0, 0.0, 0.0]), dtype=np.float64)
        assert self.base_pose.shape == (3,)
        if config["operating_points"]["enabled"]:
            self.use_operating_points = True
        bal = config["balancing"]
        self.balancing_settings.enabled = bal["enabled"]
        self.balancing_settings.arrangement_name = bal["arrangement"]
        self.balancing_settings.force_weight = bal["force_weight"]
        if bodies is None or contacts is None:
            # wrappers.py:303-305: core.parsing.parse_control_objects(config)
            from .arrangement import parse_control_objects

            bodies, contacts = parse_control_objects(config)
        self.balancing_settings.bodies = bodies
        self.balancing_settings.contacts = contacts
        if self.balancing_settings.enabled:
            self.dims.c = len(contacts)
            self.dims.nf = 1 if bal["frictionless"] else 3
        else:
            self.dims.c = 0
            self.dims.nf = 0
        ia = config["inertial_alignment"]
        self.inertial_alignment_settings.cost_enabled = ia["cost_enabled"]
        self.inertial_alignment_settings.constraint_enabled = ia["constraint_enabled"]
        # wrappers.py:347-376
        obs = config["obstacles"]
        self.obstacle_settings.enabled = obs["enabled"]
        if self.obstacle_settings.enabled:
            for pair in obs.get("collision_pairs") or []:
                self.obstacle_settings.collision_link_pairs.push_back(tuple(pair))
            self.obstacle_settings.minimum_distance = obs["minimum_distance"]
            # obstacle geometry: the reference compiles obstacles.urdf from xacro includes; here the include names
            # select the sphere table (upright_amd/robots.py documents the numbers and their xacro provenance)
            self.obstacle_settings.obstacle_urdf_path = ";".join((obs.get("urdf") or {}).get("includes", []))
            x0_obs = []
            for oc in obs.get("dynamic") or []:   # wrappers.py:363-383
                o = bindings.DynamicObstacle()
                o.name, o.radius = oc["name"], oc["radius"]
                for mc in oc["modes"]:
                    m = bindings.DynamicObstacleMode()
                    m.time = mc["time"]; m.position = np.array(mc["position"], dtype=np.float64)
                    m.velocity = np.array(mc["velocity"], dtype=np.float64); m.acceleration = np.array(mc["acceleration"], dtype=np.float64)
                    o.modes.push_back(m)
                self.obstacle_settings.dynamic_obstacles.push_back(o)
                x0_obs.append(np.concatenate([o.modes[0].position, np.zeros(6)]))   # static until the sensors update it
            self.dims.o = len(x0_obs)
            self._x0_obs = np.concatenate(x0_obs) if x0_obs else np.zeros(0)
        if x0 is None:
            x0 = pa(config["robot"]["x0"])
            assert x0.shape == (self.dims.robot.x,)
            x0 = np.concatenate([x0, getattr(self, "_x0_obs", np.zeros(0))])
        self.initial_state = np.array(x0, dtype=np.float64)
        assert self.initial_state.shape == (self.dims.x(),)
        self.xd = np.array(config.get("desired_state", np.zeros_like(self.initial_state)), dtype=np.float64)


def objects_from_fixture(arr):
    """(bodies dict, contacts list) of bindings objects from one entry of tests/golden/arrangements.json."""
    from .core_bindings import ContactPoint, RigidBody

    bodies = {b["name"]: RigidBody.from_parameters(b["params"]) for b in arr["bodies"]}
    contacts = []
    for c in arr["contacts"]:
        p = ContactPoint()
        p.object1_name, p.object2_name, p.mu = c["object1_name"], c["object2_name"], c["mu"]
        p.normal, p.span = np.array(c["normal"]), np.array(c["span"])
        p.r_co_o1, p.r_co_o2 = np.array(c["r_co_o1"]), np.array(c["r_co_o2"])
        contacts.append(p)
    return bodies, contacts


class ReplanSchedule:
    """When to re-solve: at most once per `period` of controller time, counted from the last solve
    (`tracking.min_policy_update_time`, controller.yaml:33; the cadence of manager.py:158-168).  Also the log of
    when each re-solve happened and how long it took in wall-clock seconds."""

    def __init__(self, period):
        self.period = float(period)
        self.last = -np.inf
        self.times = []
        self.durations = []

    def due(self, t):
        return t >= self.last + self.period

    def start_at(self, t):
        """A solve at time t that is not logged (the warm start)."""
        self.last = t

    def timed(self, t, solve):
        """Run `solve()`, log it as the re-solve of time t."""
        tic = time.time()
        solve()
        self.durations.append(time.time() - tic)
        self.times.append(t)
        self.last = t


class StateInputTrajectory:
    """What `plan()` returns (upright_control trajectory.py): times, states and inputs of a rolled-out plan."""

    def __init__(self, ts, xs, us):
        self.ts, self.xs, self.us = np.array(ts), np.array(xs), np.array(us)


class ChainRobot:
    """The part of robot.py's `PinocchioRobot` the callers of the controller use (mpc_sim.py:76,184; manager.py:31-97,
    140-150): `forward_xu`, then the tool link's pose / velocity / classical acceleration in the world frame -- on the serial
    chain of upright_amd/robots.py (Pinocchio and the URDF are absent here)."""

    def __init__(self, settings):
        self.dims = settings.dims
        self.chain = robots.from_config({"base_type": bindings.robot_base_type_to_string(settings.robot_base_type),
                                         "dims": {"q": settings.dims.robot.q}, "base_pose": list(settings.base_pose)})
        self._m = None

    def forward_xu(self, x, u=None):
        d = self.dims.robot
        x = np.asarray(x, dtype=np.float64)
        self._m = self.chain.forward_motion(x[: d.q], x[d.q: d.q + d.v], x[d.q + d.v: d.q + 2 * d.v])

    forward = forward_xu

    def link_pose(self, rotation_matrix=False):
        p, C = self._m[0], self._m[1]
        return (p, C) if rotation_matrix else (p, rot_to_quat_xyzw(C))

    def link_velocity(self):
        return self._m[2], self._m[3]

    def link_classical_acceleration(self):
        return self._m[4], self._m[5]


class ControllerModel:
    """manager.py:14-97: the settings plus a kinematic model of the robot for the caller's own logging."""

    def __init__(self, settings):
        self.settings = settings
        self.robot, self.geom = ChainRobot(settings), None

    @classmethod
    def from_config(cls, config, x0=None, bodies=None, contacts=None):
        return cls(ControllerSettings(config, x0=x0, bodies=bodies, contacts=contacts))

    def update(self, x, u=None):
        self.robot.forward_xu(x, u)

    def angle_between_acc_and_normal(self):
        """manager.py:66-86: angle between the tray normal and the total (inertial + gravitational) acceleration."""
        C_we = self.robot.link_pose(rotation_matrix=True)[1]
        a_ew_w, _ = self.robot.link_classical_acceleration()
        total = a_ew_w - np.asarray(self.settings.gravity)
        return np.arccos(C_we[:, 2] @ (total / np.linalg.norm(total)))

    def ddC_we_norm(self):
        """manager.py:88-97: spectral norm of the second time derivative of the tray's rotation matrix."""
        C_we = self.robot.link_pose(rotation_matrix=True)[1]
        w = self.robot.link_velocity()[1]
        al = self.robot.link_classical_acceleration()[1]
        S = lambda v: np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])   # noqa: E731
        return np.linalg.norm((S(al) + S(w) @ S(w)) @ C_we, ord=2)


class ControllerManager:
    """One controller in closed loop (the role of manager.py:100-209): owns a `bindings.ControllerInterface`, feeds
    it every observation, re-solves on the schedule above and hands back the policy's state / input for the tick.
    Attribute and method names are the reference's, so mpc_sim.py-style loops run unchanged; the first argument is a
    `ControllerModel` as in the reference (a bare `ControllerSettings` is wrapped into one)."""

    def __init__(self, model, ref_trajectory, timestep):
        self.model = model if hasattr(model, "settings") else ControllerModel(model)
        settings = self.settings = self.model.settings
        self.ref = ref_trajectory
        self.timestep = timestep
        self.schedule = ReplanSchedule(timestep)
        self.mpc = bindings.ControllerInterface(settings)
        self.mpc.reset(ref_trajectory)
        # caller-visible buffers, mutated in place by evaluateMpcSolution (manager.py:115-116,172)
        self.x_opt = np.zeros(settings.dims.x())
        self.u_opt = np.zeros(settings.dims.u())

    # the reference exposes these three as plain attributes
    last_planning_time = property(lambda self: self.schedule.last)
    replanning_times = property(lambda self: self.schedule.times)
    replanning_durations = property(lambda self: self.schedule.durations)

    @classmethod
    def from_config(cls, config, x0=None, bodies=None, contacts=None):
        """Controller dict -> manager whose target is the configured waypoints relative to the end-effector pose at
        the initial state (wrappers.py:31-43)."""
        model = ControllerModel.from_config(config, x0=x0, bodies=bodies, contacts=contacts)
        model.update(x=model.settings.initial_state)          # manager.py:131-133
        r_ew_w, Q_we = model.robot.link_pose()
        target = TargetTrajectories.from_config(config, r_ew_w, Q_we, np.zeros(model.settings.dims.u()))
        return cls(model, target, config["tracking"]["min_policy_update_time"])

    def update(self, ref):
        self.ref = ref
        self.mpc.reset(ref)

    def warmstart(self):
        self.mpc.setObservation(0, self.settings.initial_state, np.zeros(self.settings.dims.u()))
        self.mpc.advanceMpc()
        self.schedule.start_at(0)

    def step(self, t, x):
        self.mpc.setObservation(t, x, self.u_opt)
        if self.schedule.due(t):
            self.schedule.timed(t, self.mpc.advanceMpc)
        self.mpc.evaluateMpcSolution(t, x, self.x_opt, self.u_opt)
        return self.x_opt, self.u_opt

    def get_mpc_trajectory(self):
        ts, xs, us = bindings.scalar_array(), bindings.vector_array(), bindings.vector_array()
        self.mpc.getMpcSolution(ts, xs, us)
        return np.array(ts), np.array(xs), np.array(us)

    def plan(self, timestep, duration):
        """Roll the closed loop forward on the plan itself: every `timestep` the policy's own state becomes the next
        observation (manager.py:186-209)."""
        ts, xs, us = [], [], []
        t, x = 0.0, self.settings.initial_state
        while t <= duration:
            x, u = self.step(t, x)
            ts.append(t); xs.append(x.copy()); us.append(u.copy())
            t += timestep
        return StateInputTrajectory(ts, xs, us)


class BatchControllerManager:
    """B controllers of one problem family in lock step on one GPU: what a sweep over start states, targets or
    inertial-parameter samples runs (planning_sim_loop.py:613-655 does them one after the other).  Same cadence and
    call order as `ControllerManager`, every call batched: observations x[B][nx], outputs x_opt[B][nx], u_opt[B][nu].

    settings: a `ControllerSettings`; x0[B][nx] start states; targets[B][n_way][3] end-effector target positions
    (default: every instance's own end-effector position at x0 plus the configured waypoint offsets);
    body_params[B][nb][10] per-instance inertial parameters (default: the arrangement's)."""

    def __init__(self, settings, config, x0, targets=None, body_params=None):
        from .engine import BatchMPC

        self.settings = settings
        self.problem = bindings.problem_from_settings(settings)
        x0 = np.atleast_2d(np.asarray(x0, dtype=np.float64))
        self.B = x0.shape[0]
        self.x0 = x0
        way = config["waypoints"]
        self.problem.way_t = np.array([w["time"] for w in way], dtype=np.float64)
        if targets is None:
            chain = self.problem.chain
            offs = np.array([w["position"] for w in way], dtype=np.float64)
            targets = np.stack([chain.forward(x[: self.problem.nq])[0] + offs for x in x0])
        self.problem.way_p = np.asarray(targets[0], dtype=np.float64).reshape(len(way), 3)
        # target orientations per instance: Q_EE(x0) (x) Q_offset of every waypoint (wrappers.py:31-43); they enter the cost only
        # with non-zero orientation weights (weights.end_effector[3:6])
        chain = self.problem.chain
        way_q = np.stack([[quat_multiply_xyzw(rot_to_quat_xyzw(chain.forward(x[: self.problem.nq])[1]),
                                              np.asarray(w.get("orientation", [0, 0, 0, 1]), dtype=np.float64)) for w in way] for x in x0])
        self.problem.way_q = way_q[0]
        self.mpc = BatchMPC(self.problem, self.B, body_params=body_params, way_p=np.asarray(targets, dtype=np.float64), way_q=way_q)
        self.schedule = ReplanSchedule(config["tracking"]["min_policy_update_time"])
        self.x_opt = np.zeros((self.B, self.problem.nx_full))
        self.u_opt = np.zeros((self.B, self.problem.nu))
        self._fresh = True   # no previous solution yet: the next solve runs init_sqp_iteration iterations

    @classmethod
    def from_config(cls, config, x0, targets=None, body_params=None, bodies=None, contacts=None):
        x0 = np.atleast_2d(np.asarray(x0, dtype=np.float64))
        settings = ControllerSettings(config, x0=x0[0], bodies=bodies, contacts=contacts)
        return cls(settings, config, x0, targets=targets, body_params=body_params)

    def _solve(self, t, x, tick=False):
        cold = bool(self.settings.mpc.cold_start)
        if cold and not self._fresh:
            self.mpc.reset()
        if self._fresh or cold:
            self.mpc.set_sqp_iterations(int(self.settings.sqp.init_sqp_iteration))
        out = None
        if tick:    # observation in, solve, policy at the observation out: one call, one synchronisation (upr_batch_tick)
            out = self.mpc.tick(t, x)
        else:
            self.mpc.set_observation(t, x)
            self.mpc.advance()
        self._fresh = False
        return out

    def warmstart(self):
        self._solve(0.0, self.x0)
        self.schedule.start_at(0)

    def step(self, t, x):
        x = np.asarray(x, dtype=np.float64).reshape(self.B, self.problem.nx_full)
        if self.schedule.due(t):
            res = []
            self.schedule.timed(t, lambda: res.append(self._solve(t, x, tick=True)))
            xo, uo = res[0]
        else:
            xo, uo = self.mpc.evaluate(t, x if self.problem.use_feedback_policy else None)
        self.x_opt[:], self.u_opt[:] = xo, uo
        return self.x_opt, self.u_opt

    def qp_status(self):
        """Status of the last solve's final QP per instance: 0 solved, 1 iteration limit, 2 numerical failure."""
        return self.mpc.stats()["qp_status_last"].astype(int)

    def get_mpc_trajectory(self):
        ts, xs, us = self.mpc.solution()
        return ts, xs, us
