"""ctypes view of the C-ABI in include/upright_mi.h (libupright_mi.so).

This is the binding a maintainer of the reference would use in place of the pybind11 module
`upright_control.bindings` / `upright_core.bindings`.  The library is compiled from
upright_amd/csrc/*.hip by `__graft_entry__.build()`; if it is missing, importing the product fails
loudly -- there is no Python or CPU fallback.
"""
import ctypes as C
import os
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
# UPR_LIB selects another build of the same library for debugging (tools/build_prof.sh: libupright_mi_prof.so); the
# default, and what the tests, smoke() and bench.py load, is the production build
LIB_PATH = HERE / os.environ.get("UPR_LIB", "libupright_mi.so")

MAXJ, MAXC, MAXB, MAXW, MAXNX, MAXNU, MAXS, MAXP = 12, 32, 8, 8, 36, 108, 16, 32
NSTATS = 12
STAT_NAMES = (
    "sqp_iters_done", "qp_iters_last", "qp_status_last", "step_alpha_last", "cost", "constraint_violation",
    "qp_res_stat", "qp_res_eq", "qp_res_ineq", "qp_res_comp", "dx_norm", "du_norm",
)
d = C.c_double
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


class UprProblem(C.Structure):
    _fields_ = [
        ("nq", C.c_int), ("nb", C.c_int), ("nc", C.c_int), ("nf", C.c_int), ("N", C.c_int), ("dt", d),
        ("joint_type", C.c_int * MAXJ), ("joint_axis", d * 3 * MAXJ), ("joint_R", d * 9 * MAXJ), ("joint_p", d * 3 * MAXJ),
        ("tool_R", d * 9), ("tool_p", d * 3), ("gravity", d * 3),
        ("contact_body1", C.c_int * MAXC), ("contact_body2", C.c_int * MAXC), ("contact_mu", d * MAXC),
        ("contact_normal", d * 3 * MAXC), ("contact_span", d * 6 * MAXC), ("contact_r1", d * 3 * MAXC), ("contact_r2", d * 3 * MAXC),
        ("Qdiag", d * MAXNX), ("Rdiag", d * MAXNU), ("xd", d * MAXNX), ("Wee", d * 6),
        ("x_lb", d * MAXNX), ("x_ub", d * MAXNX), ("u_lb", d * MAXNU), ("u_ub", d * MAXNU),
        ("n_way", C.c_int), ("way_t", d * MAXW),
        ("sqp_iters", C.c_int), ("qp_iter_max", C.c_int), ("qp_tol", d), ("delta_tol", d), ("cost_tol", d),
        ("terminal_constraint", C.c_int),
        ("use_feedback_policy", C.c_int),
        ("n_sph", C.c_int), ("sph_frame", C.c_int * MAXS), ("sph_off", (C.c_double * 3) * MAXS), ("sph_r", C.c_double * MAXS),
        ("n_pairs", C.c_int), ("pair_a", C.c_int * MAXP), ("pair_b", C.c_int * MAXP), ("obs_min_dist", C.c_double),
        ("n_dyn", C.c_int), ("n_proj", C.c_int), ("proj_sph", C.c_int * 8), ("proj_dist", C.c_double * 8), ("proj_scale", C.c_double),
        ("soft_state_box", C.c_int), ("soft_input_box", C.c_int), ("soft_poly", C.c_int),
        ("soft_L2_lower", d), ("soft_L2_upper", d), ("soft_L1_lower", d), ("soft_L1_upper", d),
        ("soft_eq", C.c_int),
        ("qp_tol_stat", d),
    ]


def _fill(arr, values):
    flat = np.asarray(values, dtype=np.float64).ravel()
    np.ctypeslib.as_array(arr).reshape(-1)[: flat.size] = flat


def problem_to_c(P):
    """upright_amd.problem.Problem -> UprProblem (shared part; body parameters and waypoint positions
    are per instance and passed separately)."""
    P.validate()
    if P.nq > MAXJ or P.nc > MAXC or P.nb > MAXB or len(P.way_t) > MAXW:
        raise ValueError("problem exceeds the compiled-in maxima of libupright_mi")
    o = UprProblem()
    o.nq, o.nb, o.nc, o.nf, o.N, o.dt = P.nq, P.nb, P.nc, P.nf, P.N, P.dt
    for i, j in enumerate(P.chain.joints):
        o.joint_type[i] = j.kind
        _fill(o.joint_axis[i], j.axis)
        _fill(o.joint_R[i], j.R)
        _fill(o.joint_p[i], j.p)
    _fill(o.tool_R, P.chain.tool_R)
    _fill(o.tool_p, P.chain.tool_p)
    _fill(o.gravity, P.gravity)
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
    o.sqp_iters, o.qp_iter_max, o.qp_tol = int(P.sqp_iters), int(P.qp_iter_max), float(P.qp_tol)
    o.delta_tol, o.cost_tol, o.terminal_constraint = float(P.delta_tol), float(P.cost_tol), int(bool(P.terminal_constraint))
    o.use_feedback_policy = int(bool(getattr(P, "use_feedback_policy", False)))
    ns, npair = len(P.sph_r), len(P.pair_a)
    if ns > MAXS or npair > MAXP:
        raise ValueError(f"collision model too large for libupright_mi ({ns} spheres / {npair} pairs; max {MAXS} / {MAXP})")
    o.n_sph, o.n_pairs, o.obs_min_dist = ns, npair, float(P.obs_min_dist)
    for i in range(ns):
        o.sph_frame[i] = int(P.sph_frame[i]); o.sph_r[i] = float(P.sph_r[i]); _fill(o.sph_off[i], P.sph_off[i])
    for i in range(npair):
        o.pair_a[i], o.pair_b[i] = int(P.pair_a[i]), int(P.pair_b[i])
    o.n_dyn, o.n_proj, o.proj_scale = int(P.n_dyn), len(P.proj_sph), float(P.proj_scale)
    if o.n_proj > 8:
        raise ValueError("at most 8 projectile-path rows")
    for i in range(o.n_proj):
        o.proj_sph[i] = int(P.proj_sph[i]); o.proj_dist[i] = float(P.proj_dist[i])
    sl = P.slacks or {}
    o.soft_state_box, o.soft_input_box, o.soft_poly = int(bool(sl.get("state_box"))), int(bool(sl.get("input_box"))), int(bool(sl.get("poly_ineq")))
    o.soft_L2_lower, o.soft_L2_upper = float(sl.get("lower_L2_penalty", 100.0)), float(sl.get("upper_L2_penalty", 100.0))
    o.soft_L1_lower, o.soft_L1_upper = float(sl.get("lower_L1_penalty", 0.0)), float(sl.get("upper_L1_penalty", 0.0))
    # the object-dynamics equality reaches HPIPM as a general constraint with lg = ug: `poly_ineq` softens it too
    o.soft_eq = int(bool(sl.get("equality", sl.get("poly_ineq"))))
    o.qp_tol_stat = float(getattr(P, "qp_tol_stat", 0.0) or 0.0)
    return o


# every symbol include/upright_mi.h declares: (name, restype, argtypes)
PROTOTYPES = [
    ("upr_last_error", C.c_char_p, []),
    ("upr_device_available", C.c_int, []),
    ("upr_core_object_dynamics", C.c_int, [C.POINTER(UprProblem), dp, C.c_int, dp, dp, dp, dp, dp, dp]),
    ("upr_core_friction_rows", C.c_int, [C.POINTER(UprProblem), C.c_int, dp, dp]),
    ("upr_batch_create", C.c_void_p, [C.POINTER(UprProblem), C.c_int, dp, dp]),
    ("upr_batch_destroy", None, [C.c_void_p]),
    ("upr_batch_reset", C.c_int, [C.c_void_p, dp]),
    ("upr_batch_set_target_orientations", C.c_int, [C.c_void_p, dp]),
    ("upr_batch_set_observation", C.c_int, [C.c_void_p, dp, C.c_int, dp]),
    ("upr_batch_set_guess", C.c_int, [C.c_void_p, dp, dp]),
    ("upr_batch_advance", C.c_int, [C.c_void_p]),
    ("upr_batch_set_sqp_iterations", C.c_int, [C.c_void_p, C.c_int]),
    ("upr_batch_advance_async", C.c_int, [C.c_void_p]),
    ("upr_batch_sync", C.c_int, [C.c_void_p]),
    ("upr_batch_get_solution", C.c_int, [C.c_void_p, dp, dp, dp]),
    ("upr_batch_evaluate", C.c_int, [C.c_void_p, dp, C.c_int, dp, dp]),
    ("upr_batch_evaluate_policy", C.c_int, [C.c_void_p, dp, C.c_int, dp, dp, dp]),
    ("upr_batch_tick", C.c_int, [C.c_void_p, dp, C.c_int, dp, dp, dp, dp]),
    ("upr_batch_tick_graph_replays", C.c_longlong, [C.c_void_p]),
    ("upr_batch_get_feedback", C.c_int, [C.c_void_p, dp]),
    ("upr_batch_last_solve_ms", C.c_double, [C.c_void_p]),
    ("upr_batch_get_stats", C.c_int, [C.c_void_p, dp]),
    ("upr_batch_hold_stats", C.c_int, [C.c_void_p, C.c_int]),
    ("upr_batch_linearize_points", C.c_int, [C.c_void_p, C.c_int, ip, dp, dp, dp, dp, dp, dp, dp, dp, dp]),
    ("upr_batch_set_projectile_flag", C.c_int, [C.c_void_p, dp]),
    ("upr_batch_obstacle_rows", C.c_int, [C.c_void_p, C.c_int, dp, dp, dp]),
    ("upr_batch_eq_input_jacobian", C.c_int, [C.c_void_p, C.c_int, dp]),
    ("upr_batch_qp_step", C.c_int, [C.c_void_p, dp, dp]),
    ("upr_batch_qp_kkt", C.c_int, [C.c_void_p, dp, dp, dp, dp, dp, dp, ip]),
    ("upr_batch_qp_slacks", C.c_int, [C.c_void_p, dp]),
    ("upr_batch_device_ptrs", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("upr_batch_kernel_times", C.c_int, [C.c_void_p, dp, ip]),
    ("upr_batch_enable_timing", C.c_int, [C.c_void_p, C.c_int]),
    ("upr_batch_qp_kernel_name", C.c_char_p, [C.c_void_p]),
    ("upr_batch_copy_solution_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("upr_batch_copy_policy_device", C.c_int, [C.c_void_p, C.c_void_p]),
    ("upr_set_device", C.c_int, [C.c_int]),
    ("upr_batch_device", C.c_int, [C.c_void_p]),
    ("upr_batch_ws_doubles", C.c_longlong, [C.c_void_p]),
    ("upr_batch_reset_async", C.c_int, [C.c_void_p]),
    ("upr_batch_stream", C.c_void_p, [C.c_void_p]),
    ("upr_batch_qp_profile", C.c_int, [C.c_void_p, dp]),
    ("upr_batch_get_lin", C.c_int, [C.c_void_p, dp, ip]),
]

_lib = None


def lib():
    """Load libupright_mi.so; raise if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). upright_amd has no CPU fallback."
            )
        L = C.CDLL(str(LIB_PATH))
        for name, res, args in PROTOTYPES:
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError(lib().upr_last_error().decode() or "libupright_mi call failed")


def ptr(a):
    return a.ctypes.data_as(dp) if a is not None else None


def iptr(a):
    return a.ctypes.data_as(ip) if a is not None else None


def cont(a, dtype=np.float64):
    return np.ascontiguousarray(a, dtype=dtype)
