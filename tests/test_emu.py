"""CPU tests of the HIP kernel SOURCE through the test-only host emulation (tests/emu/upr_emu.cpp: one
thread per workgroup, same headers as the GPU build) against the oracle.  They validate index arithmetic
and the mathematics of every kernel body before any GPU time is spent; the GPU execution itself is
checked by tests/test_gpu_parity.py."""
import ctypes as C
import os
from pathlib import Path

import numpy as np
import pytest

from oracle.oracle import Oracle
from upright_amd import _capi
from upright_amd.problem import thing_problem
from upright_amd.sampling import level_tray_states, stationary_guess, waypoints_for

EMU = Path(os.environ.get("UPR_EMU_LIB", str(Path(__file__).resolve().parent / "emu" / "libupr_emu.so")))
p = _capi.ptr


class Emu:
    def __init__(self, P, B):
        self.P, self.B = P, B
        self.E = C.CDLL(str(EMU))
        for f in ("emu_qp2", "emu_qp3"):
            getattr(self.E, f).restype = C.c_long
        self.cp = _capi.problem_to_c(P)
        d = (C.c_int * 16)()
        self.E.emu_dims(C.byref(self.cp), d)
        (self.nx, self.nu, self.ne, self.np_, self.lin_stride, self.ws_stride, self.ws_dx, self.ws_du, self.lin_g, self.lin_gx,
         self.lin_cost, self.lin_grad, self.lin_hess, self.nfc) = list(d)[:14]
        self.bp = np.ascontiguousarray(np.broadcast_to(P.body_params, (B,) + P.body_params.shape))
        self.Df = np.zeros((B, self.ne, self.nfc))
        self.E.emu_make_Df(C.byref(self.cp), B, p(self.bp), p(self.Df))

    def linearize(self, way, t0, xs, us):
        lin = np.zeros((self.B, self.P.N + 1, self.lin_stride))
        self.E.emu_linearize(C.byref(self.cp), self.B, p(self.bp), p(way), p(t0), p(xs), p(us), p(lin))
        return lin

    def qp(self, kernel, xs, us, x0, lin):
        stats = np.zeros((self.B, 12))
        if kernel == 1:
            ws = np.full((self.B, self.ws_stride), np.nan)   # device memory is not zero either
            self.E.emu_qp(C.byref(self.cp), self.B, p(xs), p(us), p(x0), p(lin), p(self.Df), p(ws), p(stats))
        else:
            f = self.E.emu_qp2 if kernel == 2 else self.E.emu_qp3
            need = f(C.byref(self.cp), self.B, None, None, None, None, None, None, C.c_long(0), None)
            assert need > 0
            ws = np.zeros((self.B, need))
            assert f(C.byref(self.cp), self.B, p(xs), p(us), p(x0), p(lin), p(self.Df), p(ws), C.c_long(need), p(stats)) == 0
        n1, N = self.P.N + 1, self.P.N
        dx = ws[:, :n1 * self.nx].reshape(self.B, n1, self.nx)
        du = ws[:, n1 * self.nx:n1 * self.nx + N * self.nu].reshape(self.B, N, self.nu)
        return dx, du, stats, ws

    def linesearch(self, xs, us, x0, t0, way, lin, ws, stats):
        done = np.zeros(self.B, dtype=np.int32)
        xs, us = xs.copy(), us.copy()
        self.E.emu_linesearch(C.byref(self.cp), self.B, p(xs), p(us), p(x0), p(t0), p(self.bp), p(way), p(lin), p(ws), C.c_long(ws.shape[1]), p(stats),
                              done.ctypes.data_as(C.POINTER(C.c_int)), 0)
        return xs, us, done


def _case(arrangements, B, seed, **kw):
    P = thing_problem(arrangements["pink_bottle"], **kw)
    x0 = level_tray_states(B, seed=seed)
    way = waypoints_for(P, x0)
    xs, us = stationary_guess(x0, P.N, P.nu)
    return P, x0, way, np.ascontiguousarray(xs), np.ascontiguousarray(us)


def test_linearize_kernel_source(arrangements):
    B = 3
    P, x0, way, xs, us = _case(arrangements, B, 2)
    rng = np.random.default_rng(0)
    xs = xs + rng.uniform(-0.1, 0.1, xs.shape); us = rng.uniform(-1, 1, us.shape)
    e = Emu(P, B)
    lin = e.linearize(way, np.zeros(B), xs, us)
    for b in range(B):
        P.way_p = way[b]
        O = Oracle(P)
        for k in range(P.N):
            r = lin[b, k]
            g, gx, _ = O.eq_constraint(xs[b, k], us[b, k])
            assert np.abs(r[e.lin_g:e.lin_g + 6] - g).max() < 1e-13 and np.abs(r[e.lin_gx:e.lin_gx + 162].reshape(6, 27) - gx).max() < 1e-12
            c, cgx, _, H, _ = O.stage_cost(0.1 * k, xs[b, k], us[b, k])
            cj = 0.5 * np.sum(P.Qdiag * xs[b, k] ** 2) + 0.5 * np.sum(P.Rdiag * us[b, k] ** 2)
            assert abs(r[e.lin_cost] - (c - cj)) < 1e-13
            assert np.abs(r[e.lin_grad:e.lin_grad + 9] - (cgx - P.Qdiag * xs[b, k])[:9]).max() < 1e-13
            Hm = np.zeros((9, 9)); Hm[np.triu_indices(9)] = r[e.lin_hess:e.lin_hess + 45]; Hm = Hm + np.triu(Hm, 1).T
            assert np.abs(Hm - (H - np.diag(P.Qdiag))[:9, :9]).max() < 1e-13
        cN, CN = O.terminal_constraint(0.1 * P.N, xs[b, P.N])
        r = lin[b, P.N]
        assert np.abs(r[e.lin_grad:e.lin_grad + 3] - cN[:3]).max() < 1e-13
        assert np.abs(r[e.lin_hess:e.lin_hess + 27].reshape(3, 9) + CN[:3, :9]).max() < 1e-13
        gu = O.eq_constraint(xs[b, 0], us[b, 0])[2]
        assert np.abs(e.Df[b] - gu[:, 9:]).max() < 1e-15


@pytest.mark.parametrize("name", ["pink_bottle", "blue_cups", "foam_die2", "robust_8corner", "collision_rows"])
def test_linearize_job_form_equals_the_phase_form(arrangements, name):
    """Shapes without orientation cost run the lane jobs of upr_linearize2.h on the device (a tangent class per pass, the
    residual's tangent written out per class, the collision spheres placed by the walk lane) instead of upr_linearize.h's phases
    (dual numbers in every lane, link frames kept for the spheres).  Both sources on the same random points, record by record:
    one body with four contacts, seven cups (star), two stacked dice, the eight-corner robust arrangement, and the bottle with a
    collision model (world spheres, self-collision pairs; the whole record incl. the rows and their gradients); terminal records
    included."""
    B = 2
    if name == "collision_rows":
        P, x0, way, _, _ = _obstacle_case(arrangements, B, 5)
    else:
        P = thing_problem(arrangements[name])
        x0 = level_tray_states(B, seed=5)
        way = waypoints_for(P, x0, offset=(-0.5, 0.5, 0.0))
    xs, us = stationary_guess(x0, P.N, P.nu)
    rng = np.random.default_rng(3)
    xs = np.ascontiguousarray(xs + rng.uniform(-0.3, 0.3, xs.shape)); us = np.ascontiguousarray(rng.uniform(-1, 1, us.shape))
    e = Emu(P, B)
    try:
        e.E.emu_set_lin_form(0); phases = e.linearize(way, np.zeros(B), xs, us)
        e.E.emu_set_lin_form(1); jobs = e.linearize(way, np.zeros(B), xs, us)
    finally:
        e.E.emu_set_lin_form(1)
    used = e.lin_stride if name == "collision_rows" else e.lin_hess + 45   # (g, gx, cost, gradient, Hessian; + the rows' area where there are rows)
    assert np.abs(phases[..., :used]).max() > 1.0
    scale = np.maximum(1.0, np.abs(phases[..., :used]))
    assert (np.abs(jobs[..., :used] - phases[..., :used]) / scale).max() < 1e-12


@pytest.mark.parametrize("kernel", [1, 2, 3])
def test_qp_kernel_source(arrangements, kernel):
    """All three QP kernel structures follow the oracle's iterate path: identical steps after a fixed
    number of IPM iterations (tol = 0), and at convergence the same iteration count and the same minimiser
    up to the ball the 1e-8 KKT tolerance allows (the last iterations are ill-conditioned: weights 1e8+)."""
    B = 3
    P, x0, way, xs, us = _case(arrangements, B, 11, qp_tol=0.0, qp_iter_max=7)
    e = Emu(P, B)
    lin = e.linearize(way, np.zeros(B), xs, us)
    dx, du, stats, ws = e.qp(kernel, xs, us, x0, lin)
    for b in range(B):
        P.way_p = way[b]
        dxo, duo, so, rc = Oracle(P).qp_step(0.0, x0[b], xs[b], us[b])
        assert stats[b, 1] == 7 == so.qp_iters_last
        assert np.abs(dx[b] - dxo).max() < 1e-7 * max(1, np.abs(dxo).max())
        assert np.abs(du[b] - duo).max() < 1e-7 * max(1, np.abs(duo).max())
        assert np.allclose(stats[b, 6:10], list(so.qp_res), rtol=5e-2, atol=1e-9)
    P, x0, way, xs, us = _case(arrangements, B, 11)
    e = Emu(P, B)
    dx, du, stats, ws = e.qp(kernel, xs, us, x0, lin)
    for b in range(B):
        P.way_p = way[b]
        dxo, duo, so, rc = Oracle(P).qp_step(0.0, x0[b], xs[b], us[b])
        assert rc == 0 and stats[b, 2] == 0 and stats[b, 1] == so.qp_iters_last
        assert np.abs(dx[b] - dxo).max() < 2e-5 * max(1, np.abs(dxo).max())
        assert np.abs(du[b] - duo).max() < 2e-5 * max(1, np.abs(duo).max())
        assert max(stats[b, 6:10]) < P.qp_tol


def test_qp_kernels_agree_on_short_horizon_without_terminal_constraint(arrangements):
    B = 2
    P, x0, way, xs, us = _case(arrangements, B, 5, N=6, terminal_constraint=False)
    e = Emu(P, B)
    lin = e.linearize(way, np.zeros(B), xs, us)
    d1 = e.qp(1, xs, us, x0, lin); d2 = e.qp(2, xs, us, x0, lin)
    assert np.abs(d1[0] - d2[0]).max() < 2e-5 and np.abs(d1[1] - d2[1]).max() < 2e-4
    for b in range(B):
        P.way_p = way[b]
        dxo, duo, so, rc = Oracle(P).qp_step(0.0, x0[b], xs[b], us[b])
        assert np.abs(d1[0][b] - dxo).max() < 2e-5 and np.abs(d1[1][b] - duo).max() < 2e-4


def test_linesearch_kernel_source_and_full_step(arrangements):
    B = 3
    P, x0, way, xs, us = _case(arrangements, B, 21)
    e = Emu(P, B)
    t0 = np.zeros(B)
    lin = e.linearize(way, t0, xs, us)
    dx, du, stats, ws = e.qp(3, xs, us, x0, lin)
    xs2, us2, done = e.linesearch(xs, us, x0, t0, way, lin, ws, stats)
    for b in range(B):
        P.way_p = way[b]
        xo, uo, so, rc = Oracle(P).solve(0.0, x0[b], xs[b], us[b])
        assert np.abs(xs2[b] - xo).max() < 2e-5 and np.abs(us2[b] - uo).max() < 2e-4
        assert stats[b, 3] == so.step_alpha_last and abs(stats[b, 4] - so.cost) < 1e-7 and abs(stats[b, 5] - so.constraint_violation) < 1e-6


def _obstacle_case(arrangements, B, seed, **kw):
    """Thing + bottle with a small collision model: a world sphere right in front of the mobile base (the base has
    to give way while the arm still brings the tray to the target), the two self-collision pairs of
    obstacles/simple.yaml:37-41 and a far tray-vs-obstacle pair."""
    from upright_amd import robots

    P, x0, way, xs, us = _case(arrangements, B, seed, **kw)
    pairs = [("base_collision_link", "obs2"), ("wrist1_collision_link", "shoulder_collision_link"),
             ("wrist1_collision_link", "base_collision_link"), ("balanced_object_collision_link", "obs3")]
    cm = robots.collision_model(P.chain, pairs, spheres={"obs2": ("world", (0.0, 1.0, 0.25), 0.25), "obs3": ("world", (-0.3, 2.9, 0.9), 0.25)})
    for k, v in cm.items():
        setattr(P, k, v)
    way = waypoints_for(P, x0, offset=(1.0, 0.0, 0.0))
    return P, x0, way, xs, us


def test_collision_rows_kernel_source(arrangements):
    """sphere-pair distance rows of the linearisation kernel (values and d/dq through the same
    forward-mode lanes) against the oracle, the generic QP kernel with these state-polytopic rows against the
    oracle's IPM on the same iterate path, and the rows in the line-search merit."""
    B = 2
    P, x0, way, xs, us = _obstacle_case(arrangements, B, 4, qp_tol=0.0, qp_iter_max=6)
    rng = np.random.default_rng(1)
    e = Emu(P, B)
    assert e.lin_stride == 223 + 4 * 10
    xr = xs + rng.uniform(-0.2, 0.2, xs.shape)
    lin = e.linearize(way, np.zeros(B), xr, us)
    O = Oracle(P)
    for b in range(B):
        for k in range(P.N):
            d, dq = O.obstacle_rows(xr[b, k])
            r = lin[b, k, 223:]
            assert np.abs(r[:4] - d).max() < 1e-13 and np.abs(r[4:].reshape(4, 9) - dq).max() < 1e-12
    # the rows decide the step: same iterate path as the oracle after 6 IPM iterations
    lin = e.linearize(way, np.zeros(B), xs, us)
    dx, du, stats, ws = e.qp(1, xs, us, x0, lin)
    for b in range(B):
        P.way_p = way[b]
        dxo, duo, so, rc = Oracle(P).qp_step(0.0, x0[b], xs[b], us[b])
        assert stats[b, 1] == 6 == so.qp_iters_last
        assert np.abs(dx[b] - dxo).max() < 1e-6 * max(1, np.abs(dxo).max())
        assert np.abs(du[b] - duo).max() < 1e-6 * max(1, np.abs(duo).max())
    # the production kernel (rows folded into the end-effector Hessian slot of its matrix sweep) follows the same path
    dx3, du3, stats3, _ = e.qp(3, xs, us, x0, lin)
    assert np.all(stats3[:, 1] == 6)
    assert np.abs(dx3 - dx).max() < 1e-7 * max(1, np.abs(dx).max()) and np.abs(du3 - du).max() < 1e-7 * max(1, np.abs(du).max())
    # ... and they matter: without them the base drives into the obstacle's margin
    P2, _, _, _, _ = _case(arrangements, B, 4, qp_tol=0.0, qp_iter_max=6)
    e2 = Emu(P2, B)
    dx2 = e2.qp(1, xs, us, x0, e2.linearize(way, np.zeros(B), xs, us))[0]
    assert np.abs(dx2 - dx).max() > 1e-2
    # line-search merit sees a violated row
    xs_bad = xs.copy(); xs_bad[:, 5, 0] += 0.6          # base 0.6 m towards the obstacle at knot 5
    lin_bad = e.linearize(way, np.zeros(B), xs_bad, us)
    ws0 = np.zeros_like(ws); st0 = np.zeros_like(stats)
    _, _, _ = e.linesearch(xs_bad, us, x0, np.zeros(B), way, lin_bad, ws0, st0)
    for b in range(B):
        P.way_p = way[b]
        perf = Oracle(P).performance(0.0, x0[b], xs_bad[b], us[b])
        assert perf[3] > 1e-3 and abs(st0[b, 5] - np.sqrt(perf[1] + perf[2] + perf[3])) < 1e-10


def _projectile_case(arrangements, B, **kw):
    """Thing + bottle, a ball (dynamic obstacle, obstacles/dynamic.yaml:5-17) whose path crosses the straight tray
    path 0.6 m from the start one second from now; rows: two self-collision pairs, wrist-vs-ground, tray-vs-ball and
    one projectile-path row on the tray's collision link (ral23/experiments/projectile/_base.yaml:81-87)."""
    from upright_amd import robots
    from upright_amd.problem import THING_HOME

    P = thing_problem(arrangements["pink_bottle"], **kw)
    pairs = [("wrist1_collision_link_0", "shoulder_collision_link_0"), ("wrist3_collision_link_0", "ground"),
             ("forearm_collision_sphere_link2_0", "projectile1")]
    for k, v in robots.collision_model(P.chain, pairs, dynamic={"projectile1": 0.2}).items():
        setattr(P, k, v)
    P.n_dyn = 1
    robots.add_projectile_rows(P, ["balanced_object_collision_link"], [0.35], 0.2)
    x0 = np.tile(np.concatenate([THING_HOME, np.zeros(18)]), (B, 1))
    p, _ = P.chain.forward(THING_HOME)
    way = np.tile(p + np.array([0.0, -1.2, 0.0]), (B, 1, 1))
    T = 1.0
    v0 = np.array([2.5, 0.0, 0.5 * 9.81 * T]); a0 = np.array([0.0, 0.0, -9.81])
    dyn = np.zeros((B, 9))
    for b in range(B):
        cross = p + np.array([0.0, -0.6 - 0.05 * b, 0.25])
        dyn[b] = np.concatenate([cross - v0 * T - 0.5 * a0 * T * T, v0, a0])
    xs, us = stationary_guess(x0, P.N, P.nu)
    return P, x0, way, np.ascontiguousarray(xs), np.ascontiguousarray(us), dyn


def test_projectile_rows_kernel_source(arrangements):
    """dynamic-obstacle sphere, ground half-space and projectile-path rows of the linearisation kernel
    against the oracle (the obstacle is propagated ballistically to every knot; the projectile row's gradient holds
    the closest time fixed as projectile_path_constraint.h:118-145 does), and the QP on the oracle's iterate path."""
    B = 2
    P, x0, way, xs, us, dyn = _projectile_case(arrangements, B, qp_tol=0.0, qp_iter_max=4)
    assert list(P.sph_frame).count(-2) == 1 and list(P.pair_b).count(-1) == 1 and len(P.proj_sph) == 1
    e = Emu(P, B)
    flags = np.array([1.0, 1.0])
    e.E.emu_set_dynamic(p(dyn), p(flags))
    rng = np.random.default_rng(3)
    xr = xs + rng.uniform(-0.2, 0.2, xs.shape)
    lin = e.linearize(way, np.zeros(B), xr, us)
    nrow = 4
    assert e.lin_stride == 223 + nrow * 10
    for b in range(B):
        O = Oracle(P); O.set_dynamic_obstacle(dyn[b], 1.0)
        for k in range(P.N):
            d, dq = O.obstacle_rows(xr[b, k], tau=k * P.dt)
            r = lin[b, k, 223:]
            # collision rows to rounding; the projectile row goes through a Newton iteration with a 1e-4 stopping test
            assert np.abs(r[:3] - d[:3]).max() < 1e-12 and np.abs(r[nrow:].reshape(nrow, 9)[:3] - dq[:3]).max() < 1e-11
            assert abs(r[3] - d[3]) < 1e-9 and np.abs(r[nrow:].reshape(nrow, 9)[3] - dq[3]).max() < 1e-8
    lin = e.linearize(way, np.zeros(B), xs, us)
    dx, du, stats, ws = e.qp(1, xs, us, x0, lin)
    for b in range(B):
        P.way_p = way[b]
        O = Oracle(P); O.set_dynamic_obstacle(dyn[b], 1.0)
        dxo, duo, so, rc = O.qp_step(0.0, x0[b], xs[b], us[b])
        # (this first QP is infeasible -- the goal lies behind the linearised rows -- so the iterates part quickly)
        assert np.abs(dx[b] - dxo).max() < 1e-5 * max(1, np.abs(dxo).max())
    dx3, du3, stats3, _ = e.qp(3, xs, us, x0, lin)
    assert np.all(stats3[:, 1] == stats[:, 1]) and np.abs(dx3 - dx).max() < 1e-5 * max(1, np.abs(dx).max())
    e.E.emu_set_dynamic(None, None)


def test_two_dynamic_obstacles_kernel_source(arrangements):
    """More than one dynamic obstacle (dimensions.h:32-45; system_pinocchio_mapping.h:84-97): spheres with sph_frame -2 - i ride
    on obstacle i, every obstacle is propagated ballistically to every knot, and the projectile-path row follows the LAST
    obstacle (projectile_path_constraint.h:82: state.tail(9)).  Linearisation kernel source against the oracle."""
    from upright_amd import robots

    B = 2
    P, x0, way, xs, us, ball = _projectile_case(arrangements, B)
    pairs = [("wrist3_collision_link_0", "ground"), ("base_collision_link_0", "chair1"), ("forearm_collision_sphere_link2_0", "projectile1"),
             ("balanced_object_collision_link_0", "chair1")]
    for k, v in robots.collision_model(P.chain, pairs, dynamic={"chair1": 0.25, "projectile1": 0.2}).items():
        setattr(P, k, v)
    P.n_dyn = 2
    P.proj_sph = np.zeros(0, dtype=np.int32); P.proj_dist = np.zeros(0)
    robots.add_projectile_rows(P, ["balanced_object_collision_link"], [0.35], 0.2)
    assert sorted(f for f in P.sph_frame if f <= -2) == [-3, -2]
    chair = np.tile(np.array([0.3, 0.4, 0.25, 0.0, 0.2, 0.0, 0.05, 0.0, 0.0]), (B, 1)); chair[1, :3] += 0.2
    dyn = np.ascontiguousarray(np.concatenate([chair, ball], axis=1))       # [B][2][9]: chair first, the ball last
    e = Emu(P, B)
    flags = np.ones(B)                       # (kept alive: the emulation holds the pointer)
    e.E.emu_set_dynamic(p(dyn), p(flags))
    rng = np.random.default_rng(5)
    xr = xs + rng.uniform(-0.2, 0.2, xs.shape)
    lin = e.linearize(way, np.zeros(B), xr, us)
    nrow = 5
    assert e.lin_stride == 223 + nrow * 10
    for b in range(B):
        O = Oracle(P); O.set_dynamic_obstacle(dyn[b], 1.0)
        for k in range(P.N):
            d, dq = O.obstacle_rows(xr[b, k], tau=k * P.dt)
            r = lin[b, k, 223:]
            assert np.abs(r[:4] - d[:4]).max() < 1e-12 and np.abs(r[nrow:].reshape(nrow, 9)[:4] - dq[:4]).max() < 1e-11
            assert abs(r[4] - d[4]) < 1e-9 and np.abs(r[nrow:].reshape(nrow, 9)[4] - dq[4]).max() < 1e-8
    e.E.emu_set_dynamic(None, None)


def test_soft_rows_kernel_source(arrangements):
    """HPIPM slack variables (hpipm_interface SlackSettings, wrappers.py:121-143): the generic QP kernel with
    softened state-box / input-box / polytopic rows follows the oracle's iterate path, on a feasible instance and on
    one whose first knot is infeasible (base acceleration beyond what the friction cone balances) where the hard QP
    ends at its iteration cap."""
    B = 2
    soft = dict(state_box=True, input_box=True, poly_ineq=True, equality=False, lower_L2_penalty=100.0, upper_L2_penalty=50.0, lower_L1_penalty=0.0, upper_L1_penalty=0.5)
    P, x0, way, xs, us = _case(arrangements, B, 11, qp_tol=0.0, qp_iter_max=6)
    x0[1, 18] = 5.0
    xs[1, :, 18] = 5.0
    P.slacks = soft
    e = Emu(P, B)
    lin = e.linearize(way, np.zeros(B), xs, us)
    dx, du, stats, ws = e.qp(1, xs, us, x0, lin)
    for b in range(B):
        P.way_p = way[b]
        dxo, duo, so, rc = Oracle(P).qp_step(0.0, x0[b], xs[b], us[b])
        assert stats[b, 1] == 6 == so.qp_iters_last
        assert np.abs(dx[b] - dxo).max() < 1e-7 * max(1, np.abs(dxo).max())
        assert np.abs(du[b] - duo).max() < 1e-7 * max(1, np.abs(duo).max())
        assert np.allclose(stats[b, 6:10], list(so.qp_res), rtol=5e-2, atol=1e-9)
    # (the violated instance stalls near 1e-8 in r_stat / r_eq -- weights span 16 decades -- so converge to 1e-7)
    P.qp_tol, P.qp_iter_max = 1e-7, 40
    e = Emu(P, B)
    dx, du, stats, ws = e.qp(1, xs, us, x0, lin)
    for b in range(B):
        P.way_p = way[b]
        dxo, duo, so, rc = Oracle(P).qp_step(0.0, x0[b], xs[b], us[b])
        assert rc == 0 and stats[b, 2] == 0 and abs(stats[b, 1] - so.qp_iters_last) <= 1
        assert np.abs(dx[b] - dxo).max() < 2e-5 * max(1, np.abs(dxo).max())
        assert np.abs(du[b] - duo).max() < 2e-5 * max(1, np.abs(duo).max())
    # the production kernel's source with the same slacks (SOFT instantiation: boxes in registers, friction rows in its far arrays)
    dx3, du3, stats3, _ = e.qp(3, xs, us, x0, lin)
    for b in range(B):
        assert stats3[b, 2] == 0 and abs(stats3[b, 1] - stats[b, 1]) <= 1
        assert np.abs(dx3[b] - dx[b]).max() < 2e-5 * max(1, np.abs(dx[b]).max()) and np.abs(du3[b] - du[b]).max() < 2e-5 * max(1, np.abs(du[b]).max())
    # only the polytopic rows softened: boxes stay hard (instance 1 then violates its acceleration box: not compared)
    P.slacks = dict(poly_ineq=True, equality=False)
    e = Emu(P, B)
    for kernel in (1, 3):
        dx, du, stats, ws = e.qp(kernel, xs, us, x0, lin)
        for b in range(1):
            P.way_p = way[b]
            dxo, duo, so, rc = Oracle(P).qp_step(0.0, x0[b], xs[b], us[b])
            assert stats[b, 2] == 0 and rc == 0
            assert np.abs(dx[b] - dxo).max() < 2e-5 * max(1, np.abs(dxo).max())
            assert np.abs(du[b] - duo).max() < 2e-5 * max(1, np.abs(duo).max())
    # fixed iteration count: the production kernel's soft friction rows follow the oracle's iterate path
    P.qp_tol, P.qp_iter_max = 0.0, 6
    e = Emu(P, B)
    dx, du, stats, ws = e.qp(3, xs, us, x0, lin)
    P.way_p = way[0]
    dxo, duo, so, rc = Oracle(P).qp_step(0.0, x0[0], xs[0], us[0])
    assert stats[0, 1] == 6 == so.qp_iters_last
    assert np.abs(dx[0] - dxo).max() < 1e-7 * max(1, np.abs(dxo).max()) and np.abs(du[0] - duo).max() < 1e-7 * max(1, np.abs(duo).max())


def _emu_kkt(e, ws, b):
    """Multipliers the generic kernel left in the instance workspace, in the layout of BatchMPC.qp_kkt()."""
    o = (C.c_int * 8)()
    e.E.emu_kkt_offsets(C.byref(e.cp), o)
    ws_pi, ws_nu, ws_yN, ws_lam, ni, neN = list(o)[:6]
    n1, N = e.P.N + 1, e.P.N
    w = ws[b]
    lam = w[ws_lam:ws_lam + n1 * ni].reshape(n1, ni).copy()
    lam[0, :2 * e.nx] = 0.0                       # slots that are not rows of the stage
    lam[N, 2 * e.nx:] = 0.0
    if ni > 2 * e.nx + 2 * e.nu + e.np_:
        lam[0, 2 * e.nx + 2 * e.nu + e.np_:] = 0.0
    return dict(pi=w[ws_pi:ws_pi + n1 * e.nx].reshape(n1, e.nx), nu=w[ws_nu:ws_nu + N * e.ne].reshape(N, e.ne),
                yN=w[ws_yN:ws_yN + neN], lam=lam)


def test_kkt_conditions_checked_in_numpy(arrangements):
    """Optimality of the QP kernel's primal-dual point verified WITHOUT the oracle: tests/kkt_check.py assembles the
    sub-problem in numpy from the linearisation records and the problem constants and evaluates stationarity,
    feasibility and complementarity at what the kernel returned (N = 20, headline shape; then the softened 8-body
    problem of BASELINE config 4).  The same check runs on the GPU kernels in tests/test_gpu_parity.py."""
    from kkt_check import force_jacobian, kkt_residuals

    B = 2
    P, x0, way, xs, us = _case(arrangements, B, 5, qp_tol=1e-9, qp_iter_max=40)
    e = Emu(P, B)
    assert np.abs(force_jacobian(P, P.body_params) - e.Df[0]).max() < 1e-15
    lin = e.linearize(way, np.zeros(B), xs, us)
    dx, du, stats, ws = e.qp(1, xs, us, x0, lin)
    assert np.all(stats[:, 2] == 0)
    for b in range(B):
        sol = dict(dx=dx[b], du=du[b], **_emu_kkt(e, ws, b))
        res = kkt_residuals(P, P.body_params, x0[b], xs[b], us[b], lin[b], sol)
        assert res.max() < 1e-7, res
        # the check has teeth: a perturbed primal point or multiplier is not a KKT point
        bad = dict(sol); bad["du"] = du[b].copy(); bad["du"][3, 2] += 1e-3
        assert kkt_residuals(P, P.body_params, x0[b], xs[b], us[b], lin[b], bad).max() > 1e-6
        bad = dict(sol); bad["lam"] = sol["lam"].copy(); bad["lam"][5, 2 * P.nx + 1] += 1e-3
        assert kkt_residuals(P, P.body_params, x0[b], xs[b], us[b], lin[b], bad).max() > 1e-6
    # BASELINE config 4: eight bodies, frictionless, HPIPM slacks on the state boxes and the (equality) general rows
    from upright_amd.problem import THING_HOME

    P = thing_problem(arrangements["robust_8corner"], nf=1, force_weight=0.0, qp_tol=1e-9, qp_iter_max=40)
    P.slacks = dict(state_box=True, input_box=False, poly_ineq=True)
    x0 = np.tile(np.concatenate([THING_HOME, np.zeros(18)]), (1, 1))
    way = waypoints_for(P, x0, offset=(-2.0, 1.0, 0.0))
    xs, us = stationary_guess(x0, P.N, P.nu); xs = np.ascontiguousarray(xs); us = np.ascontiguousarray(us)
    e = Emu(P, 1)
    lin = e.linearize(way, np.zeros(1), xs, us)
    dx, du, stats, ws = e.qp(1, xs, us, x0, lin)
    assert stats[0, 2] == 0
    sol = dict(dx=dx[0], du=du[0], **_emu_kkt(e, ws, 0))
    res = kkt_residuals(P, P.body_params, x0[0], xs[0], us[0], lin[0], sol)
    assert res.max() < 1e-7, res


def test_end_effector_orientation_cost_kernel_source(arrangements):
    """cost/end_effector_cost.h:33-84 with orientation weights and reference_trajectory.h:18-47 with two waypoints: the
    linearisation kernel's orientation error (rotation-matrix form, tangents by the dual-number lanes) against the
    oracle's quaternion form (ocs2 quaternionDistance of the extracted quaternion, SLERP'd target): cost, gradient and
    Gauss-Newton Hessian at times inside, at and outside the waypoint interval; then an MPC solve with the line search."""
    B = 2
    P, x0, way, xs, us = _case(arrangements, B, 9)
    P.Wee = np.array([1.0, 1.0, 1.0, 0.3, 0.5, 0.2])
    P.way_t = np.array([0.4, 1.6])
    rng = np.random.default_rng(4)
    p0 = np.stack([P.chain.forward(x[:9])[0] for x in x0])
    way = np.ascontiguousarray(np.stack([p0 + [0.2, 0.0, 0.05], p0 + [-0.4, 0.5, 0.0]], axis=1))
    q = rng.normal(size=(B, 2, 4)); q /= np.linalg.norm(q, axis=2, keepdims=True)
    from upright_amd.control import rot_to_quat_xyzw, quat_multiply_xyzw
    for b in range(B):       # targets a moderate rotation away from the current orientation (theta < pi)
        qe = rot_to_quat_xyzw(P.chain.forward(x0[b, :9])[1])
        for w in range(2):
            dq = np.concatenate([0.3 * q[b, w, :3], [1.0]]); dq /= np.linalg.norm(dq)
            q[b, w] = quat_multiply_xyzw(qe, dq)
    q = np.ascontiguousarray(q)
    e = Emu(P, B)
    e.E.emu_set_way_q(p(q))
    xs = xs + rng.uniform(-0.2, 0.2, xs.shape); us = rng.uniform(-1, 1, us.shape)
    lin = e.linearize(way, np.zeros(B), xs, us)
    nq, nh = 9, 45
    o_cost = 6 + 6 * 27
    iu = np.triu_indices(nq)
    for b in range(B):
        P.way_p, P.way_q = way[b], q[b]
        O = Oracle(P)
        for k in (0, 3, 4, 10, 16, 19):
            t = k * P.dt
            c, gx, gu, H, R = O.stage_cost(t, xs[b, k], us[b, k])
            c -= 0.5 * np.sum(P.Qdiag * (xs[b, k] - P.xd) ** 2) + 0.5 * np.sum(P.Rdiag * us[b, k] ** 2)
            gx = gx - P.Qdiag * (xs[b, k] - P.xd)
            H = H - np.diag(P.Qdiag)
            rec = lin[b, k]
            assert abs(rec[o_cost] - c) < 1e-12 * max(1.0, abs(c))
            assert np.abs(rec[o_cost + 1:o_cost + 1 + nq] - gx[:nq]).max() < 1e-11
            Hk = np.zeros((nq, nq)); Hk[iu] = rec[o_cost + 1 + nq:o_cost + 1 + nq + nh]; Hk = Hk + np.triu(Hk, 1).T
            assert np.abs(Hk - H[:nq, :nq]).max() < 1e-11
        # the orientation part is really there
        P0 = thing_problem(arrangements["pink_bottle"]); P0.way_t, P0.way_p = P.way_t, way[b]
        assert abs(Oracle(P0).stage_cost(1.0, xs[b, 10], us[b, 10])[0] - O.stage_cost(1.0, xs[b, 10], us[b, 10])[0]) > 1e-4
    # one SQP iteration (QP + line search on the kernel source) against the oracle
    xs0, us0 = stationary_guess(x0, P.N, P.nu); xs0 = np.ascontiguousarray(xs0); us0 = np.ascontiguousarray(us0)
    lin = e.linearize(way, np.zeros(B), xs0, us0)
    dx, du, stats, ws = e.qp(3, xs0, us0, x0, lin)
    xs1, us1, done = e.linesearch(xs0, us0, x0, np.zeros(B), way, lin, ws, stats)
    for b in range(B):
        P.way_p, P.way_q = way[b], q[b]
        xo, uo, so, rc = Oracle(P).solve(0.0, x0[b], xs0[b], us0[b])
        assert rc == 0 and np.abs(xs1[b] - xo).max() < 2e-5 and np.abs(us1[b] - uo).max() < 2e-4
    e.E.emu_set_way_q(None)


@pytest.mark.parametrize("horizon", [2.0, 1.0])
def test_ur10_shape_production_kernel_source(arrangements, horizon):
    """BASELINE configs[0] (ur10_demo.yaml: fixed-base UR10, nq 6, one body, four frictionless contacts: nx 18, nu 10) on the
    production QP kernel's source: the (6, 1, 4, 1) instantiations at horizon N = 20 (time_horizon 2.0, as shipped) and
    N = 10 (time_horizon 1.0, the horizon BASELINE.json names).  The four normal forces cannot span the six object-dynamics
    rows, so the hard equality gets the proximal treatment of upr_qp.h inside the kernel (until round 3 such problems ran the
    second-structure kernel).  Against the generic kernel's source and the oracle on the same linearisation records."""
    import copy
    import json
    from pathlib import Path

    from upright_amd import control

    cfg = copy.deepcopy(json.load(open(Path(__file__).parent / "golden" / "configs.json"))["ur10_demo"]["controller"])
    cfg["mpc"]["time_horizon"] = horizon
    bodies, contacts = control.objects_from_fixture(arrangements["pink_bottle"])
    settings = control.ControllerSettings(cfg, bodies=bodies, contacts=contacts)
    from upright_amd import control_bindings as cb

    P = cb.problem_from_settings(settings)
    assert (P.nq, P.nb, P.nc, P.nf, P.N) == (6, 1, 4, 1, int(round(horizon / 0.1)))
    x0 = np.array(settings.initial_state)
    # (frictionless contacts hold no tangential load: the fixed first knot is feasible only on an exactly level tray)
    _, Cm = P.chain.forward(x0[:P.nq])
    ez = np.array([0.0, 0.0, 1.0]); v = Cm.T @ ez; ax = np.cross(ez, v); s_, c_ = np.linalg.norm(ax), ez @ v
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    P.chain.tool_R = P.chain.tool_R @ (np.eye(3) + K + K @ K * ((1 - c_) / (s_ * s_)))
    P.qp_tol, P.qp_iter_max = 1e-9, 60
    p0, _ = P.chain.forward(x0[:P.nq])
    B = 2
    x0b = np.tile(x0, (B, 1))
    way = np.stack([(p0 + np.array(off))[None] for off in ([0.3, 0.2, 0.1], [-0.2, 0.3, 0.0])])
    P.way_t, P.way_p = np.zeros(1), way[0]
    xs, us = stationary_guess(x0b, P.N, P.nu)
    xs, us = np.ascontiguousarray(xs), np.ascontiguousarray(us)
    e = Emu(P, B)
    lin = e.linearize(way, np.zeros(B), xs, us)
    dx1, du1, st1, _ = e.qp(1, xs, us, x0b, lin)
    dx3, du3, st3, _ = e.qp(3, xs, us, x0b, lin)
    for b in range(B):
        P.way_p = way[b]
        dxo, duo, so, rc = Oracle(P).qp_step(0.0, x0b[b], xs[b], us[b])
        assert rc == 0 and so.qp_status_last == 0 and st1[b, 2] == 0 and st3[b, 2] == 0
        assert abs(st3[b, 1] - so.qp_iters_last) <= 1
        for dx, du in ((dx1, du1), (dx3, du3)):
            assert np.abs(dx[b] - dxo).max() < 2e-5 * max(1.0, np.abs(dxo).max())
            assert np.abs(du[b][:, :P.nq] - duo[:, :P.nq]).max() < 2e-4 * max(1.0, np.abs(duo).max())


def test_closed_form_ee_tangents_equal_the_forward_mode_walk(arrangements):
    """upr_ee_from_snap (what the linearisation kernel runs: every tangent of the end-effector state as a closed form of its
    joint's snapshot) against the forward-mode walk on (value, tangent) pairs, for all 30 entries of the state -- the
    VELOCITY included, which no term of the path consumes yet (it carried a zero tangent) -- along all 3 nq
    state coordinates (the Thing chain: two prismatic and seven revolute joints), at random states with non-zero joint rates
    and accelerations."""
    P = thing_problem(arrangements["pink_bottle"])
    E = C.CDLL(str(EMU))
    cp = _capi.problem_to_c(P)
    rng = np.random.default_rng(17)
    nx = 3 * P.nq
    for _ in range(5):
        x = rng.uniform(-1.0, 1.0, nx)
        a = np.zeros((nx, 2, 30)); b = np.zeros((nx, 2, 30))
        E.emu_ee_tangents(C.byref(cp), p(x), p(a), p(b))
        scale = max(1.0, np.abs(b).max())
        assert np.abs(a[:, 0] - b[:, 0]).max() < 1e-13 * scale
        assert np.abs(a[:, 1] - b[:, 1]).max() < 1e-12 * scale, np.unravel_index(np.abs(a[:, 1] - b[:, 1]).argmax(), (nx, 30))
        assert np.abs(b[:, 1, 12:15]).max() > 0.1      # the velocity tangent is not trivially zero


@pytest.mark.parametrize("name", ["blue_cups", "foam_die2"])
def test_paper_arrangement_shapes_production_kernel_source(arrangements, name):
    """The production kernel's instantiations for the paper's other arrangements -- seven cups (star arrangement WITH friction,
    nu = 93: compact Df, no staged Z, far hf / ek: upr_qp3_cfg::BIGF) and two stacked dice (dense 12 x 12 Schur complement) --
    follow the oracle's iterate path: identical steps after a fixed number of interior-point iterations."""
    B = 2
    P = thing_problem(arrangements[name], qp_tol=0.0, qp_iter_max=6)
    x0 = level_tray_states(B, seed=13)
    way = waypoints_for(P, x0, offset=(-0.5, 0.5, 0.0))
    xs, us = stationary_guess(x0, P.N, P.nu)
    xs, us = np.ascontiguousarray(xs), np.ascontiguousarray(us)
    e = Emu(P, B)
    lin = e.linearize(way, np.zeros(B), xs, us)
    dx, du, stats, ws = e.qp(3, xs, us, x0, lin)
    for b in range(B):
        P.way_p = way[b]
        dxo, duo, so, rc = Oracle(P).qp_step(0.0, x0[b], xs[b], us[b])
        assert stats[b, 1] == 6 == so.qp_iters_last
        assert np.abs(dx[b] - dxo).max() < 1e-7 * max(1, np.abs(dxo).max())
        assert np.abs(du[b] - duo).max() < 1e-7 * max(1, np.abs(duo).max())
