"""Batched MPC engine: thin Python owner of a `upr_batch` handle (include/upright_mi.h).

One `BatchMPC` = B independent instances of one problem family on the current HIP device.  It is what
`bindings.ControllerInterface` wraps with B = 1, and what the benchmark / multi-GPU driver shard over
ranks.  All numerics run in libupright_mi.so; this file only moves arrays across the C-ABI.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import check, cont, iptr, ptr


class BatchMPC:
    def __init__(self, problem, B=1, body_params=None, way_p=None, way_q=None, device=None):
        self.problem = problem.validate()
        self.B = int(B)
        self.nx, self.nu, self.N = problem.nx, problem.nu, problem.N
        self.nxf = problem.nx_full   # interface state: robot state + 9 per dynamic obstacle
        self.ne = 6 * problem.nb
        if body_params is None:
            body_params = np.broadcast_to(problem.body_params, (self.B,) + problem.body_params.shape)
        if way_p is None:
            way_p = np.broadcast_to(problem.way_p, (self.B,) + problem.way_p.shape)
        self.body_params = cont(body_params).reshape(self.B, problem.nb, 10)
        self.way_p = cont(way_p).reshape(self.B, len(problem.way_t), 3)
        self._c = _capi.problem_to_c(problem)
        self._lib = _capi.lib()
        if device is not None:     # one process per GPU: the launcher's LOCAL_RANK (None: the thread's current device)
            check(self._lib.upr_set_device(int(device)))
        self._h = self._lib.upr_batch_create(C.byref(self._c), self.B, ptr(self.body_params), ptr(self.way_p))
        if not self._h:
            raise RuntimeError(self._lib.upr_last_error().decode())
        if way_q is None and getattr(problem, "way_q", None) is not None:
            way_q = np.broadcast_to(np.asarray(problem.way_q, dtype=np.float64), (self.B, len(problem.way_t), 4))
        if way_q is not None:
            self.set_target_orientations(way_q)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.upr_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- ControllerInterface surface, batched ------------------------------------------------------
    def reset(self, way_p=None):
        if way_p is not None:
            self.way_p = cont(way_p).reshape(self.B, len(self.problem.way_t), 3)
        check(self._lib.upr_batch_reset(self._h, ptr(self.way_p) if way_p is not None else None))

    def set_target_orientations(self, way_q):
        """Target orientations per instance and waypoint, (B, n_way, 4) quaternions xyzw."""
        self.way_q = cont(way_q).reshape(self.B, len(self.problem.way_t), 4)
        check(self._lib.upr_batch_set_target_orientations(self._h, ptr(self.way_q)))

    def set_observation(self, t, x):
        x = cont(x).reshape(self.B, self.nxf)
        t = cont(np.broadcast_to(np.asarray(t, dtype=np.float64), (self.B,)))
        check(self._lib.upr_batch_set_observation(self._h, ptr(t), 1, ptr(x)))

    def set_guess(self, xs, us):
        xs = cont(xs).reshape(self.B, self.N + 1, self.nxf)
        us = cont(us).reshape(self.B, self.N, self.nu)
        check(self._lib.upr_batch_set_guess(self._h, ptr(xs), ptr(us)))

    def set_sqp_iterations(self, n):
        """SQP iterations of the next advance only (sqp.init_sqp_iteration of a solve without a previous solution)."""
        check(self._lib.upr_batch_set_sqp_iterations(self._h, int(n)))

    def advance(self):
        check(self._lib.upr_batch_advance(self._h))

    def advance_async(self):
        check(self._lib.upr_batch_advance_async(self._h))

    def sync(self):
        check(self._lib.upr_batch_sync(self._h))

    def solution(self):
        ts = np.zeros((self.B, self.N + 1))
        xs = np.zeros((self.B, self.N + 1, self.nxf))
        us = np.zeros((self.B, self.N, self.nu))
        check(self._lib.upr_batch_get_solution(self._h, ptr(ts), ptr(xs), ptr(us)))
        return ts, xs, us

    def tick(self, t, x, want_stats=False):
        """One control period: set_observation(t, x), advance(), evaluate(t, x_obs=x) in one call and one synchronisation
        (upr_batch_tick).  Returns (x_opt, u_opt) or (x_opt, u_opt, stats)."""
        t = cont(np.broadcast_to(np.asarray(t, dtype=np.float64), (self.B,)))
        xo = cont(x).reshape(self.B, self.nxf)
        xr = np.zeros((self.B, self.nxf)); u = np.zeros((self.B, self.nu))
        s = np.zeros((self.B, _capi.NSTATS)) if want_stats else None
        check(self._lib.upr_batch_tick(self._h, ptr(t), 1, ptr(xo), ptr(xr), ptr(u), ptr(s) if want_stats else None))
        if want_stats:
            return xr, u, {name: s[:, i] for i, name in enumerate(_capi.STAT_NAMES)}
        return xr, u

    def tick_graph_replays(self):
        """Control periods upr_batch_tick served by replaying its captured HIP graph."""
        return int(self._lib.upr_batch_tick_graph_replays(self._h))

    def evaluate(self, t, x_obs=None):
        """Plan state and input at time t.  With x_obs (B, nx) and use_feedback_policy the input is the linear
        policy u*(t) + K(t) (x_obs - x*(t)) of the last solve (ocs2::LinearController), else the feed-forward input."""
        t = cont(np.broadcast_to(np.asarray(t, dtype=np.float64), (self.B,)))
        x = np.zeros((self.B, self.nxf))
        u = np.zeros((self.B, self.nu))
        if x_obs is not None and self.problem.use_feedback_policy:
            xo = cont(x_obs).reshape(self.B, self.nxf)
            check(self._lib.upr_batch_evaluate_policy(self._h, ptr(t), 1, ptr(xo), ptr(x), ptr(u)))
        else:
            check(self._lib.upr_batch_evaluate(self._h, ptr(t), 1, ptr(x), ptr(u)))
        return x, u

    def feedback_gains(self):
        """K[B][N][nu][nx] at the knots of the last solve (u = bias + K x, ocs2 sign)."""
        K = np.zeros((self.B, self.N, self.nu, self.nxf))
        check(self._lib.upr_batch_get_feedback(self._h, ptr(K)))
        return K

    def last_solve_ms(self):
        return float(self._lib.upr_batch_last_solve_ms(self._h))

    def stats(self):
        s = np.zeros((self.B, _capi.NSTATS))
        check(self._lib.upr_batch_get_stats(self._h, ptr(s)))
        return {name: s[:, i] for i, name in enumerate(_capi.STAT_NAMES)}

    def preserved_stats(self):
        """Context manager: statistics and QP dispatch keys of the last advance are put back on exit (upr_batch_hold_stats) --
        around a query that solves one more QP on the handle."""
        import contextlib

        @contextlib.contextmanager
        def hold():
            check(self._lib.upr_batch_hold_stats(self._h, 0))
            try:
                yield
            finally:
                check(self._lib.upr_batch_hold_stats(self._h, 1))

        return hold()

    # -- term-level access ----------------------------------------------------------------------------
    def linearize_points(self, x, u, t=None, inst=None):
        x = cont(x).reshape(-1, self.nxf)
        n = x.shape[0]
        u = cont(u).reshape(n, self.nu)
        t = cont(np.zeros(n) if t is None else np.broadcast_to(np.asarray(t, dtype=np.float64), (n,)))
        inst = cont(np.zeros(n) if inst is None else inst, dtype=np.int32)
        nq = self.problem.nq
        out = dict(
            g=np.zeros((n, self.ne)), gx=np.zeros((n, self.ne, self.nx)), cost=np.zeros(n),
            grad=np.zeros((n, nq)), hess=np.zeros((n, nq, nq)), ee=np.zeros((n, 3)),
        )
        check(self._lib.upr_batch_linearize_points(
            self._h, n, iptr(inst), ptr(t), ptr(x), ptr(u), ptr(out["g"]), ptr(out["gx"]), ptr(out["cost"]),
            ptr(out["grad"]), ptr(out["hess"]), ptr(out["ee"])))
        return out

    def obstacle_rows(self, x, jac=True):
        """Collision rows d (n, n_pairs) and d d / d q (n, n_pairs, nq) at n states (`obstacle_avoidance`)."""
        x = cont(x).reshape(-1, self.nxf)
        n, npair, nq = x.shape[0], len(self.problem.pair_a) + len(self.problem.proj_sph), self.problem.nq
        d = np.zeros((n, npair)); dq = np.zeros((n, npair, nq))
        check(self._lib.upr_batch_obstacle_rows(self._h, n, ptr(x), ptr(d), ptr(dq) if jac else None))
        return (d, dq) if jac else d

    def set_projectile_flag(self, s):
        """Activation flag of the projectile rows per instance (8th entry of the target state)."""
        s = cont(np.broadcast_to(np.asarray(s, dtype=np.float64), (self.B,)))
        check(self._lib.upr_batch_set_projectile_flag(self._h, ptr(s)))

    def eq_input_jacobian(self, inst=0):
        gu = np.zeros((self.ne, self.nu))
        check(self._lib.upr_batch_eq_input_jacobian(self._h, int(inst), ptr(gu)))
        return gu

    def qp_step(self):
        dxs = np.zeros((self.B, self.N + 1, self.nxf))
        dus = np.zeros((self.B, self.N, self.nu))
        check(self._lib.upr_batch_qp_step(self._h, ptr(dxs), ptr(dus)))
        return dxs, dus

    def qp_kkt(self):
        """One QP at the current trajectory: step and the multipliers the kernel ended with (see upr_batch_qp_kkt)."""
        P = self.problem
        ni = C.c_int(0)
        nin = 2 * self.nx + 2 * self.nu + (5 * P.nc if P.nf == 3 else 0) + len(P.pair_a) + len(P.proj_sph)
        out = dict(dx=np.zeros((self.B, self.N + 1, self.nx)), du=np.zeros((self.B, self.N, self.nu)),
                   pi=np.zeros((self.B, self.N + 1, self.nx)), nu=np.zeros((self.B, self.N, self.ne)),
                   yN=np.zeros((self.B, 3 + 2 * P.nq if P.terminal_constraint else 0)), lam=np.zeros((self.B, self.N + 1, nin)))
        check(self._lib.upr_batch_qp_kkt(self._h, ptr(out["dx"]), ptr(out["du"]), ptr(out["pi"]), ptr(out["nu"]),
                                         ptr(out["yN"]) if out["yN"].size else None, ptr(out["lam"]), C.byref(ni)))
        assert ni.value == nin, (ni.value, nin)
        out["slack"] = np.ones((self.B, self.N + 1, nin))   # the rows' slacks at the exit (lam / slack = barrier weights of the last iterate)
        check(self._lib.upr_batch_qp_slacks(self._h, ptr(out["slack"])))
        return out

    def device_ptrs(self):
        xs, us = C.c_void_p(), C.c_void_p()
        check(self._lib.upr_batch_device_ptrs(self._h, C.byref(xs), C.byref(us)))
        return xs.value, us.value

    def copy_solution_device(self, xs_ptr, us_ptr):
        check(self._lib.upr_batch_copy_solution_device(self._h, C.c_void_p(xs_ptr), C.c_void_p(us_ptr)))

    def copy_policy_device(self, u_ptr):
        """u_0 of every instance as the last tick() evaluated it, device -> device on the engine's stream (asynchronous)."""
        check(self._lib.upr_batch_copy_policy_device(self._h, C.c_void_p(u_ptr)))

    def device_index(self):
        return int(self._lib.upr_batch_device(self._h))

    def stream_ptr(self):
        """The engine's HIP stream (hipStream_t) as an integer, e.g. for torch.cuda.ExternalStream."""
        return int(self._lib.upr_batch_stream(self._h) or 0)

    def reset_async(self):
        check(self._lib.upr_batch_reset_async(self._h))

    def qp_profile(self):
        """Debug: arm (first call) / read-and-clear the per-phase cycle counters of the QP kernel, [B][wave 0..3][16]."""
        out = np.zeros((self.B, 4, 16))
        check(self._lib.upr_batch_qp_profile(self._h, ptr(out)))
        return out

    def lin_records(self):
        stride = C.c_int(0)
        check(self._lib.upr_batch_get_lin(self._h, None, C.byref(stride)))
        lin = np.zeros((self.B, self.N + 1, stride.value))
        check(self._lib.upr_batch_get_lin(self._h, ptr(lin), C.byref(stride)))
        return lin

    def enable_timing(self, on=True):
        """on: False / 0 no events; True / 1 around every kernel of an advance; 2 around the QP kernel only; 3 around every
        fourth QP launch."""
        check(self._lib.upr_batch_enable_timing(self._h, int(on)))

    def kernel_times(self):
        ms = np.zeros(3)
        n = np.zeros(3, dtype=np.int32)
        check(self._lib.upr_batch_kernel_times(self._h, ptr(ms), iptr(n)))
        return dict(linearize_ms=ms[0], qp_ms=ms[1], linesearch_ms=ms[2], launches=n.tolist(),
                    qp_kernel=self._lib.upr_batch_qp_kernel_name(self._h).decode(), ws_doubles=int(self._lib.upr_batch_ws_doubles(self._h)))


def core_object_dynamics(problem, body_params, forces, Cm, w, al, a):
    """upright_core.bindings.compute_object_dynamics_constraints, batched over the leading axis."""
    c = _capi.problem_to_c(problem)
    forces = cont(forces).reshape(-1, problem.nf * problem.nc)
    n = forces.shape[0]
    out = np.zeros((n, 6 * problem.nb))
    check(_capi.lib().upr_core_object_dynamics(
        C.byref(c), ptr(cont(body_params).reshape(problem.nb, 10)), n, ptr(forces), ptr(cont(Cm).reshape(n, 9)),
        ptr(cont(w).reshape(n, 3)), ptr(cont(al).reshape(n, 3)), ptr(cont(a).reshape(n, 3)), ptr(out)))
    return out


def core_friction_rows(problem, forces):
    c = _capi.problem_to_c(problem)
    forces = cont(forces).reshape(-1, 3 * problem.nc)
    n = forces.shape[0]
    out = np.zeros((n, 5 * problem.nc))
    check(_capi.lib().upr_core_friction_rows(C.byref(c), n, ptr(forces), ptr(out)))
    return out
