"""Flat, numeric description of one optimal-control problem family (shared by every instance of a batch).

This is the data model of SURVEY.md section 7.1: dimensions, kinematic chain, balanced bodies
(10 inertial parameters each), contact table, weights, bounds, target waypoints and solver settings.
It mirrors what `ControllerInterface::ControllerInterface`
(`upright_control/src/controller_interface.cpp:103-393`) reads out of `ControllerSettings`.
"""
from dataclasses import dataclass, field

import numpy as np

from . import robots


@dataclass
class Problem:
    chain: robots.Chain
    # bodies in std::map (sorted-name) order (contact_constraints.h:180)
    body_names: list
    body_params: np.ndarray  # (nb, 10)  [m, m*c, Ixx, Ixy, Ixz, Iyy, Iyz, Izz]  rigid_body.h:47-51
    # contacts (contact.h:10-46)
    contact_body1: np.ndarray  # (nc,) int, -1 = EE / fixture
    contact_body2: np.ndarray
    contact_mu: np.ndarray
    contact_normal: np.ndarray  # (nc, 3)
    contact_span: np.ndarray  # (nc, 2, 3)
    contact_r1: np.ndarray
    contact_r2: np.ndarray
    nf: int = 3
    N: int = 20
    dt: float = 0.1
    gravity: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, -9.81]))
    Qdiag: np.ndarray = None
    Rdiag: np.ndarray = None
    xd: np.ndarray = None
    Wee: np.ndarray = field(default_factory=lambda: np.array([1.0, 1, 1, 0, 0, 0]))
    x_lb: np.ndarray = None
    x_ub: np.ndarray = None
    u_lb: np.ndarray = None
    u_ub: np.ndarray = None
    way_t: np.ndarray = field(default_factory=lambda: np.zeros(1))
    way_p: np.ndarray = field(default_factory=lambda: np.zeros((1, 3)))
    way_q: np.ndarray = None  # (n_way, 4) target orientations, xyzw (None = identity); used when Wee[3:] != 0
    sqp_iters: int = 1
    qp_iter_max: int = 30
    qp_tol: float = 1e-8        # equality / inequality / complementarity residuals (HPIPM tol_eq, tol_ineq, tol_comp)
    qp_tol_stat: float = 1e-6   # stationarity residual (HPIPM tol_stat; ocs2's hpipm_interface::Settings default)
    delta_tol: float = 1e-3
    cost_tol: float = 1e-4
    terminal_constraint: bool = True
    use_feedback_policy: bool = False  # sqp.use_feedback_policy (controller.yaml:60)
    # collision avoidance (controller_interface.cpp:172-228): spheres on chain frames (-1 world, i < nq link after
    # joint i, nq tool frame) and the pairs kept obs_min_dist apart at knots 1..N-1
    sph_frame: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int32))
    sph_off: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    sph_r: np.ndarray = field(default_factory=lambda: np.zeros(0))
    pair_a: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int32))
    pair_b: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int32))
    obs_min_dist: float = 0.1  # controller.yaml:108
    # pair_b == -1: ground half-space; sph_frame == -2 - i: dynamic obstacle i (its state [r, v, a] is the i-th 9-block appended
    # to x, system_dynamics.h:29-39, dimensions.h:32-45); projectile path rows (projectile_path_constraint.h): spheres whose centres are the
    # checked link positions, their distances and the scale
    n_dyn: int = 0
    proj_sph: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int32))
    proj_dist: np.ndarray = field(default_factory=lambda: np.zeros(0))
    proj_scale: float = 1.0
    # HPIPM soft constraints (hpipm_interface SlackSettings, wrappers.py:121-143): None / {} = hard constraints; keys
    # state_box, input_box, poly_ineq (bool), lower/upper_L2_penalty (100), lower/upper_L1_penalty (0)
    slacks: dict = None

    @property
    def nq(self):
        return self.chain.nq

    @property
    def nx_full(self):
        """State dimension at the interface: robot state + 9 per dynamic obstacle (dimensions.h:32-45)."""
        return 3 * self.nq + 9 * self.n_dyn

    @property
    def nb(self):
        return len(self.body_names)

    @property
    def nc(self):
        return len(self.contact_mu)

    @property
    def nx(self):
        return 3 * self.nq

    @property
    def nu(self):
        return self.nq + self.nf * self.nc

    def validate(self):
        nx, nu = self.nx, self.nu
        for name, n in (("Qdiag", nx), ("xd", nx), ("x_lb", nx), ("x_ub", nx), ("Rdiag", nu), ("u_lb", nu), ("u_ub", nu)):
            a = np.asarray(getattr(self, name), dtype=np.float64)
            if a.shape != (n,):
                raise ValueError(f"{name} has shape {a.shape}, expected ({n},)")
        if self.nf not in (1, 3):
            raise ValueError("nf must be 1 (frictionless) or 3")
        if np.asarray(self.Wee).shape != (6,) or np.any(np.asarray(self.Wee) < 0):
            raise ValueError("Wee must hold six non-negative weights")
        if self.body_params.shape != (self.nb, 10):
            raise ValueError("body_params must be (nb, 10)")
        return self


def contacts_from_fixture(arr):
    """(names, params, contact arrays) from one entry of tests/golden/arrangements.json or from the
    output of the arrangement parser: bodies sorted by name, contacts in list order (the contact order
    fixes the layout of the force block of u, balancing_constraints.cpp:63,118)."""
    bodies = sorted(arr["bodies"], key=lambda b: b["name"])
    names = [b["name"] for b in bodies]
    params = np.array([b["params"] for b in bodies], dtype=np.float64).reshape(len(names), 10)
    idx = {n: i for i, n in enumerate(names)}
    cs = arr["contacts"]
    b1 = np.array([idx.get(c["object1_name"], -1) for c in cs], dtype=np.int32)
    b2 = np.array([idx[c["object2_name"]] for c in cs], dtype=np.int32)
    return dict(
        body_names=names,
        body_params=params,
        contact_body1=b1,
        contact_body2=b2,
        contact_mu=np.array([c["mu"] for c in cs], dtype=np.float64),
        contact_normal=np.array([c["normal"] for c in cs], dtype=np.float64).reshape(len(cs), 3),
        contact_span=np.array([c["span"] for c in cs], dtype=np.float64).reshape(len(cs), 2, 3),
        contact_r1=np.array([c["r_co_o1"] for c in cs], dtype=np.float64).reshape(len(cs), 3),
        contact_r2=np.array([c["r_co_o2"] for c in cs], dtype=np.float64).reshape(len(cs), 3),
    )


# upright_cmd/config/robots/thing.yaml:48,61-86
THING_HOME = np.array([-1.0, 1.0, 0.0, 0.5 * np.pi, -0.25 * np.pi, 0.5 * np.pi, -0.25 * np.pi, 0.5 * np.pi, 0.417 * np.pi])


def thing_problem(arrangement, nf=3, N=20, dt=0.1, force_weight=0.001, waypoint_offset=(-2.0, 1.0, 0.0), x0=None, mount_yaw=None, **kw):
    """Headline configuration H: Thing + arrangement, weights/limits of
    `robots/thing.yaml:61-86`, `controller.yaml:54-79`, target = EE(x0) + offset
    (`wrappers.py:31-43`, `ral23/experiments/_point1.yaml:3-8`)."""
    chain = robots.thing(mount_yaw=mount_yaw)
    c = contacts_from_fixture(arrangement)
    nq = 9
    nc = len(c["contact_mu"])
    nu = nq + nf * nc
    two_pi = 2 * np.pi
    x_ub = np.array([10, 10, 10] + [two_pi] * 6 + [1.1, 1.1, 2, 2, 2, 3, 3, 3, 3] + [2.5, 2.5, 1, 10, 10, 10, 10, 10, 10], dtype=np.float64)
    j_ub = np.array([20, 20, 20, 80, 80, 80, 80, 80, 80], dtype=np.float64)
    # controller_interface.cpp:330-357: forces in [0, 100] (nf = 1) or [-100, 100] (nf = 3)
    f_lb = np.full(nf * nc, 0.0 if nf == 1 else -1e2)
    f_ub = np.full(nf * nc, 1e2)
    if x0 is None:
        x0 = np.concatenate([THING_HOME, np.zeros(18)])
    p0, _ = chain.forward(x0[:nq])
    P = Problem(
        chain=chain,
        nf=nf,
        N=N,
        dt=dt,
        Qdiag=0.01 * np.concatenate([np.zeros(9), 10 * np.ones(9), np.ones(9)]),
        Rdiag=np.concatenate([0.001 * np.ones(9), force_weight * np.ones(nf * nc)]),
        xd=np.zeros(27),
        x_lb=-x_ub,
        x_ub=x_ub,
        u_lb=np.concatenate([-j_ub, f_lb]),
        u_ub=np.concatenate([j_ub, f_ub]),
        way_t=np.zeros(1),
        way_p=(p0 + np.asarray(waypoint_offset, dtype=np.float64)).reshape(1, 3),
        **c,
        **kw,
    )
    return P.validate()
