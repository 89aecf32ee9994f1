"""ctypes front-end of the CPU oracle (oracle/upright_oracle.{h,cpp}).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by upright_amd/."""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "_build" / "libupright_oracle.so"

MAXJ, MAXC, MAXB, MAXW, MAXNX, MAXNU, MAXS, MAXP = 16, 64, 16, 8, 64, 208, 32, 64
MAX_DYN = 4
d = C.c_double


class OrcProblem(C.Structure):
    _fields_ = [
        ("nq", C.c_int), ("nb", C.c_int), ("nc", C.c_int), ("nf", C.c_int), ("N", C.c_int), ("dt", d),
        ("joint_type", C.c_int * MAXJ), ("joint_axis", d * 3 * MAXJ), ("joint_R", d * 9 * MAXJ), ("joint_p", d * 3 * MAXJ),
        ("tool_R", d * 9), ("tool_p", d * 3), ("gravity", d * 3),
        ("body_params", d * 10 * MAXB),
        ("contact_body1", C.c_int * MAXC), ("contact_body2", C.c_int * MAXC), ("contact_mu", d * MAXC),
        ("contact_normal", d * 3 * MAXC), ("contact_span", d * 6 * MAXC), ("contact_r1", d * 3 * MAXC), ("contact_r2", d * 3 * MAXC),
        ("Qdiag", d * MAXNX), ("Rdiag", d * MAXNU), ("xd", d * MAXNX), ("Wee", d * 6),
        ("x_lb", d * MAXNX), ("x_ub", d * MAXNX), ("u_lb", d * MAXNU), ("u_ub", d * MAXNU),
        ("n_way", C.c_int), ("way_t", d * MAXW), ("way_p", d * 3 * MAXW), ("way_q", d * 4 * MAXW),
        ("sqp_iters", C.c_int), ("qp_iter_max", C.c_int), ("qp_tol", d), ("delta_tol", d), ("cost_tol", d),
        ("terminal_constraint", C.c_int),
        ("n_sph", C.c_int), ("sph_frame", C.c_int * MAXS), ("sph_off", d * 3 * MAXS), ("sph_r", d * MAXS),
        ("n_pairs", C.c_int), ("pair_a", C.c_int * MAXP), ("pair_b", C.c_int * MAXP), ("obs_min_dist", d),
        ("n_dyn", C.c_int), ("dyn_x0", d * (9 * MAX_DYN)), ("n_proj", C.c_int), ("proj_sph", C.c_int * 8), ("proj_dist", d * 8),
        ("proj_scale", d), ("proj_s", d),
        ("soft_state_box", C.c_int), ("soft_input_box", C.c_int), ("soft_poly", C.c_int),
        ("soft_L2_lower", d), ("soft_L2_upper", d), ("soft_L1_lower", d), ("soft_L1_upper", d),
        ("soft_eq", C.c_int),
        ("qp_tol_stat", d),
    ]


class OrcStats(C.Structure):
    _fields_ = [
        ("sqp_iters_done", C.c_int), ("qp_iters_last", C.c_int), ("qp_status_last", C.c_int), ("step_alpha_last", d),
        ("cost", d), ("constraint_violation", d), ("qp_res", d * 4), ("dx_norm", d), ("du_norm", d),
    ]


def build(force=False):
    src = [HERE / "upright_oracle.cpp", HERE / "upright_oracle.h", HERE / "Makefile"]
    libs = [LIB, HERE / "_build" / "libupright_oracle_ngam0.so"]
    if force or any(not l.exists() or any(s.stat().st_mtime > l.stat().st_mtime for s in src) for l in libs):
        subprocess.check_call(["make", "-s", "-C", str(HERE)])
    return LIB


_lib = None
_variants = {}


def lib(variant=None):
    """The oracle library; variant "ngam0": the build without the centrality safeguard (oracle/Makefile)."""
    global _lib
    if variant:
        if variant not in _variants:
            build()
            v = C.CDLL(str(HERE / "_build" / f"libupright_oracle_{variant}.so"))
            v.orc_stage_cost.restype = d
            _variants[variant] = v
        return _variants[variant]
    if _lib is None:
        build()
        _lib = C.CDLL(str(LIB))
        _lib.orc_stage_cost.restype = d
    return _lib


def _fill(arr, values):
    flat = np.asarray(values, dtype=np.float64).ravel()
    buf = np.ctypeslib.as_array(arr).reshape(-1) if not isinstance(arr, np.ndarray) else arr
    buf[: flat.size] = flat


def to_orc(P):
    """upright_amd.problem.Problem -> OrcProblem."""
    o = OrcProblem()
    o.nq, o.nb, o.nc, o.nf, o.N, o.dt = P.nq, P.nb, P.nc, P.nf, P.N, P.dt
    for i, j in enumerate(P.chain.joints):
        o.joint_type[i] = j.kind
        _fill(o.joint_axis[i], j.axis)
        _fill(o.joint_R[i], j.R)
        _fill(o.joint_p[i], j.p)
    _fill(o.tool_R, P.chain.tool_R)
    _fill(o.tool_p, P.chain.tool_p)
    _fill(o.gravity, P.gravity)
    for b in range(P.nb):
        _fill(o.body_params[b], P.body_params[b])
    for i in range(P.nc):
        o.contact_body1[i] = int(P.contact_body1[i])
        o.contact_body2[i] = int(P.contact_body2[i])
        o.contact_mu[i] = float(P.contact_mu[i])
        _fill(o.contact_normal[i], P.contact_normal[i])
        _fill(o.contact_span[i], P.contact_span[i])
        _fill(o.contact_r1[i], P.contact_r1[i])
        _fill(o.contact_r2[i], P.contact_r2[i])
    _fill(o.Qdiag, P.Qdiag); _fill(o.Rdiag, P.Rdiag); _fill(o.xd, P.xd); _fill(o.Wee, P.Wee)
    _fill(o.x_lb, P.x_lb); _fill(o.x_ub, P.x_ub); _fill(o.u_lb, P.u_lb); _fill(o.u_ub, P.u_ub)
    o.n_way = len(P.way_t)
    _fill(o.way_t, P.way_t)
    for i in range(o.n_way):
        _fill(o.way_p[i], P.way_p[i])
        wq = getattr(P, "way_q", None)
        _fill(o.way_q[i], wq[i] if wq is not None else [0.0, 0.0, 0.0, 1.0])
    o.sqp_iters, o.qp_iter_max, o.qp_tol = P.sqp_iters, P.qp_iter_max, P.qp_tol
    o.delta_tol, o.cost_tol, o.terminal_constraint = P.delta_tol, P.cost_tol, int(P.terminal_constraint)
    ns, npair = len(getattr(P, "sph_r", ())), len(getattr(P, "pair_a", ()))
    o.n_sph, o.n_pairs, o.obs_min_dist = ns, npair, float(getattr(P, "obs_min_dist", 0.0))
    for i in range(ns):
        o.sph_frame[i] = int(P.sph_frame[i]); o.sph_r[i] = float(P.sph_r[i]); _fill(o.sph_off[i], P.sph_off[i])
    for i in range(npair):
        o.pair_a[i], o.pair_b[i] = int(P.pair_a[i]), int(P.pair_b[i])
    o.n_dyn = int(getattr(P, "n_dyn", 0))
    nproj = len(getattr(P, "proj_sph", ()))
    o.n_proj, o.proj_scale, o.proj_s = nproj, float(getattr(P, "proj_scale", 1.0)), 0.0
    for i in range(nproj):
        o.proj_sph[i] = int(P.proj_sph[i]); o.proj_dist[i] = float(P.proj_dist[i])
    sl = getattr(P, "slacks", None) or {}
    o.soft_state_box, o.soft_input_box, o.soft_poly = int(bool(sl.get("state_box"))), int(bool(sl.get("input_box"))), int(bool(sl.get("poly_ineq")))
    o.soft_L2_lower, o.soft_L2_upper = float(sl.get("lower_L2_penalty", 100.0)), float(sl.get("upper_L2_penalty", 100.0))
    o.soft_L1_lower, o.soft_L1_upper = float(sl.get("lower_L1_penalty", 0.0)), float(sl.get("upper_L1_penalty", 0.0))
    o.soft_eq = int(bool(sl.get("equality", sl.get("poly_ineq"))))
    o.qp_tol_stat = float(getattr(P, "qp_tol_stat", 0.0) or 0.0)
    return o


def _p(a):
    return a.ctypes.data_as(C.POINTER(d))


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Oracle:
    def __init__(self, P, variant=None):
        self.P = P
        self.o = to_orc(P)
        self.L = lib(variant)
        self.nx, self.nu = P.nx, P.nu

    def object_dynamics(self, forces, Cm, w, al, a):
        out = np.zeros(6 * self.P.nb)
        self.L.orc_object_dynamics(C.byref(self.o), _p(_c(forces)), _p(_c(Cm)), _p(_c(w)), _p(_c(al)), _p(_c(a)), _p(out))
        return out

    def friction_rows(self, forces):
        out = np.zeros(5 * self.P.nc)
        self.L.orc_friction_rows(C.byref(self.o), _p(_c(forces)), _p(out))
        return out

    def ee_kinematics(self, x, jac=False):
        out = np.zeros(24)
        dout = np.zeros((24, self.nx)) if jac else None
        self.L.orc_ee_kinematics(C.byref(self.o), _p(_c(x)), _p(out), _p(dout) if jac else None)
        return (out, dout) if jac else out

    def eq_constraint(self, x, u, jac=True):
        ne = 6 * self.P.nb
        g = np.zeros(ne)
        gx = np.zeros((ne, self.nx)) if jac else None
        gu = np.zeros((ne, self.nu)) if jac else None
        self.L.orc_eq_constraint(C.byref(self.o), _p(_c(x)), _p(_c(u)), _p(g), _p(gx) if jac else None, _p(gu) if jac else None)
        return (g, gx, gu) if jac else g

    def ineq_constraint(self, u):
        h = np.zeros(5 * self.P.nc)
        self.L.orc_ineq_constraint(C.byref(self.o), _p(_c(u)), _p(h))
        return h

    def stage_cost(self, t, x, u, derivs=True):
        gx = np.zeros(self.nx); gu = np.zeros(self.nu); H = np.zeros((self.nx, self.nx)); R = np.zeros(self.nu)
        if derivs:
            c = self.L.orc_stage_cost(C.byref(self.o), d(t), _p(_c(x)), _p(_c(u)), _p(gx), _p(gu), _p(H), _p(R))
            return c, gx, gu, H, R
        return self.L.orc_stage_cost(C.byref(self.o), d(t), _p(_c(x)), _p(_c(u)), None, None, None, None)

    def terminal_constraint(self, t, x):
        n = 3 + 2 * self.P.nq
        c = np.zeros(n); cx = np.zeros((n, self.nx))
        self.L.orc_terminal_constraint(C.byref(self.o), d(t), _p(_c(x)), _p(c), _p(cx))
        return c, cx

    def dynamics(self, x, u):
        xn = np.zeros(self.nx)
        self.L.orc_dynamics(C.byref(self.o), _p(_c(x)), _p(_c(u)), _p(xn))
        return xn

    def performance(self, t0, x0, xs, us):
        out = np.zeros(4)
        self.L.orc_performance(C.byref(self.o), d(t0), _p(_c(x0)), _p(_c(xs)), _p(_c(us)), _p(out))
        return out

    def qp_step(self, t0, x0, xs, us):
        dxs = np.zeros((self.P.N + 1, self.nx)); dus = np.zeros((self.P.N, self.nu)); st = OrcStats()
        rc = self.L.orc_qp_step(C.byref(self.o), d(t0), _p(_c(x0)), _p(_c(xs)), _p(_c(us)), _p(dxs), _p(dus), C.byref(st))
        return dxs, dus, st, rc

    def sphere_centers(self, x):
        c = np.zeros((self.o.n_sph, 3))
        self.L.orc_sphere_centers(C.byref(self.o), _p(_c(x)), _p(c))
        return c

    def set_dynamic_obstacle(self, x_obs, flag=1.0):
        """Observed states [r, v, a] of the dynamic obstacles (one after the other, 9 n_dyn values) at the start of the horizon
        and the target's activation flag of the projectile constraint."""
        x_obs = np.asarray(x_obs, dtype=np.float64).ravel()
        for i in range(9 * MAX_DYN):
            self.o.dyn_x0[i] = float(x_obs[i]) if i < len(x_obs) else 0.0
        self.o.proj_s = float(flag)

    def obstacle_rows(self, x, jac=True, tau=0.0):
        n = self.o.n_pairs + self.o.n_proj
        dd = np.zeros(n); dq = np.zeros((n, self.P.nq))
        self.L.orc_obstacle_rows(C.byref(self.o), _p(_c(x)), d(tau), _p(dd), _p(dq) if jac else None)
        return (dd, dq) if jac else dd

    def qp_feedback(self, t0, x0, xs, us):
        P = self.P
        dxs = np.zeros((P.N + 1, self.nx)); dus = np.zeros((P.N, self.nu)); K = np.zeros((P.N, self.nu, self.nx)); st = OrcStats()
        rc = self.L.orc_qp_feedback(C.byref(self.o), d(t0), _p(_c(x0)), _p(_c(xs)), _p(_c(us)), _p(dxs), _p(dus), _p(K), C.byref(st))
        return dxs, dus, K, st, rc

    def solve_batch(self, t0, x0, xs, us, way_p=None, body_params=None, nthreads=0):
        """n independent solves, OpenMP over instances inside the library.  Returns (xs, us, stats list, threads used)."""
        x0 = _c(x0); n = x0.shape[0]
        xs = _c(xs).copy(); us = _c(us).copy()
        t0 = _c(np.broadcast_to(np.asarray(t0, dtype=np.float64), (n,)))
        st = (OrcStats * n)()
        wp = _c(way_p) if way_p is not None else None
        bp = _c(body_params) if body_params is not None else None
        self.L.orc_solve_batch.restype = C.c_int
        used = self.L.orc_solve_batch(C.byref(self.o), C.c_int(n), _p(wp) if wp is not None else None, _p(bp) if bp is not None else None,
                                      _p(t0), _p(x0), _p(xs), _p(us), st, C.c_int(int(nthreads)))
        return xs, us, list(st), used

    def solve(self, t0, x0, xs, us):
        xs = _c(xs).copy(); us = _c(us).copy(); st = OrcStats()
        rc = self.L.orc_solve(C.byref(self.o), d(t0), _p(_c(x0)), _p(xs), _p(us), C.byref(st))
        return xs, us, st, rc
