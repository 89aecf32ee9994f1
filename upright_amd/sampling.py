"""Synthetic problem instances for tests and the benchmark.

Start states: the survey proposes x_home + U(-0.25, 0.25) on all nine joints.  Perturbing the wrist
joints tilts the tray by up to ~14 deg at t = 0, where the state is fixed, so the friction pyramid
(mu = 0.234 for pink_bottle) has no feasible contact force at knot 0 and the QP of those instances is
infeasible -- in the reference as well (hard constraints, `controller.yaml:70-72` slacks disabled).
The distribution used here keeps the physical premise of the task -- the tray starts level:
  * base x, y, yaw and shoulder pan (vertical axes): U(-0.25, 0.25) [m | rad]
  * shoulder lift d4, elbow d5: U(-0.25, 0.25) rad, wrist 1 = -(d4 + d5)  (pitch chain sums to zero)
  * wrist 2, wrist 3 unchanged
  * velocities: the same construction scaled to U(-0.2, 0.2); accelerations zero.
"""
import numpy as np

from .problem import THING_HOME


def level_tray_states(B, seed=0, dq=0.25, dv=0.2):
    rng = np.random.default_rng(seed)

    def level(n, s):
        d = np.zeros((n, 9))
        d[:, 0:4] = rng.uniform(-s, s, (n, 4))
        d[:, 4:6] = rng.uniform(-s, s, (n, 2))
        d[:, 6] = -(d[:, 4] + d[:, 5])
        return d

    x = np.zeros((B, 27))
    x[:, :9] = THING_HOME + level(B, dq)
    x[:, 9:18] = level(B, dv)
    return x


def contract_states(B, seed=0, dq=0.25, dv=0.2):
    """SURVEY.md section 8(d) as written: x_home + U(-dq, dq) on ALL nine joints, U(-dv, dv) on the velocities, zero
    accelerations.  69 % of these starts have no feasible contact force at the fixed first knot (see above)."""
    rng = np.random.default_rng(seed)
    x = np.zeros((B, 27))
    x[:, :9] = THING_HOME + rng.uniform(-dq, dq, (B, 9))
    x[:, 9:18] = rng.uniform(-dv, dv, (B, 9))
    return x


def stationary_guess(x0, N, nu):
    """ocs2 DefaultInitializer (controller_interface.cpp:385-386): hold the state, zero input."""
    x0 = np.atleast_2d(x0)
    xs = np.repeat(x0[:, None, :], N + 1, axis=1)
    us = np.zeros((x0.shape[0], N, nu))
    return xs, us


def waypoints_for(problem, x0, offset=(-2.0, 1.0, 0.0)):
    """Per-instance target = EE(x0) + offset (wrappers.py:31-43, ral23/experiments/_point1.yaml:3-8)."""
    x0 = np.atleast_2d(x0)
    off = np.asarray(offset, dtype=np.float64)
    return np.stack([problem.chain.forward(x[: problem.nq])[0] + off for x in x0]).reshape(x0.shape[0], 1, 3)
