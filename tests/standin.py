"""Stand-in for the engine in the multi-rank control-flow tests and `bench.py --dry-run` (no GPU here): the methods bench.py's rank
functions call, with deterministic 'solutions'.  TEST INFRASTRUCTURE: it lives under tests/ so that nothing in the product package
fabricates a solution; never used by a measurement."""
import numpy as np


class StandInEngine:
    """The methods bench.py's rank functions call on the engine, with deterministic 'solutions' (instance b's trajectory is
    filled with +-b) -- the solve itself needs a GPU.  Used by the gloo world-size-2 tests and by `bench.py --dry-run`;
    never by a measurement."""

    def __init__(self, B, lo, N, nx, nu, nxf=None, device=0):
        self.B, self.N, self.nx, self.nu, self.lo = B, N, nx, nu, lo
        self.device = int(device)
        self.nxf = nxf or nx
        ids = np.arange(lo, lo + B, dtype=np.float64)
        self.xs = np.ascontiguousarray(ids[:, None, None] + np.zeros((B, N + 1, nx)))
        self.us = np.ascontiguousarray(-ids[:, None, None] + np.zeros((B, N, nu)))
        self.calls = []
        self.t = 0.0

    def device_index(self): return self.device
    def reset_async(self): self.calls.append("reset")
    def advance_async(self): self.calls.append("advance")
    def advance(self): self.calls.append("advance")
    def sync(self): pass
    def enable_timing(self, on=True): pass
    def close(self): pass
    def set_projectile_flag(self, s): self.calls.append("flag")
    def set_observation(self, t, x): self.calls.append("obs"); self.t = float(np.max(t))

    def evaluate(self, t, x_obs=None):
        """'policy': u_0[b] = -(global id of b) - t in every component."""
        ids = np.arange(self.lo, self.lo + self.B, dtype=np.float64)
        u = -(ids[:, None] + float(np.max(t))) + np.zeros((self.B, self.nu))
        return np.zeros((self.B, self.nxf)), u

    def tick(self, t, x, want_stats=False):
        """One control period (BatchMPC.tick): observation, solve, policy at the observation."""
        self.set_observation(t, x); self.advance()
        xo, u = self.evaluate(t, x_obs=x)
        return (xo, u, self.stats()) if want_stats else (xo, u)

    def stats(self):
        return dict(qp_status_last=np.zeros(self.B), qp_iters_last=np.full(self.B, 10.0), constraint_violation=np.zeros(self.B))

    def kernel_times(self):
        return dict(linearize_ms=0.0, qp_ms=0.0, linesearch_ms=0.0, launches=[0, 0, 0], qp_kernel="stand-in")

    def copy_policy_device(self, up):
        import ctypes

        _, u = self.evaluate(self.t)
        u = np.ascontiguousarray(u)
        ctypes.memmove(up, u.ctypes.data, u.nbytes)

    def copy_solution_device(self, xp, up):
        import ctypes

        ctypes.memmove(xp, self.xs.ctypes.data, self.xs.nbytes)
        ctypes.memmove(up, self.us.ctypes.data, self.us.nbytes)


def _plain(v):
    """Settings field -> JSON-able (numbers, strings, lists, dicts), recursively over the settings classes."""
    import enum

    if isinstance(v, enum.Enum):
        return v.name
    if isinstance(v, (bool, int, float, str)) or v is None:
        return v
    if isinstance(v, (np.floating, np.integer, np.bool_)):
        return v.item()
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, dict):
        return {str(k): _plain(x) for k, x in sorted(v.items())}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if hasattr(v, "get_parameters"):          # RigidBody: its ten inertial parameters (rigid_body.h:47-51)
        return _plain(v.get_parameters())
    if hasattr(v, "__dict__"):
        return {k: _plain(x) for k, x in sorted(vars(v).items()) if not k.startswith("_")}
    raise TypeError(type(v))


# fields of ControllerSettings that are file paths of the reference's tool chain (URDF compilation, CppAD library folder):
# not inputs of the accelerated path
SETTINGS_PATH_FIELDS = ("robot_urdf_path", "lib_folder")


def dump_settings(s):
    """Every field of a `ControllerSettings` (control_bindings.py) as plain data, for the field-for-field comparison between the
    reference's wrappers.py and upright_amd/control.py (tests/golden/make_caller_fixtures.py, tests/test_host.py)."""
    d = _plain(s)
    d["dims_totals"] = {k: getattr(s.dims, k)() for k in ("q", "v", "x", "f", "u")}
    return d


class RecordingControllerInterface:
    """Stand-in for `bindings.ControllerInterface` that records every call its caller makes (name + arguments) and returns a
    deterministic 'policy': after n solves, at (t, x), x_opt = 0.5 x + n, u_opt = -n - t.  TEST INFRASTRUCTURE for the
    call-sequence comparison of manager.py against upright_amd/control.py; never a measurement."""

    def __init__(self, settings):
        self.nx, self.nu = settings.dims.x(), settings.dims.u()
        self.calls = [["ControllerInterface", self.nx, self.nu]]
        self.solves = 0

    def reset(self, target):
        self.calls.append(["reset", [float(t) for t in target.ts], [np.asarray(x).tolist() for x in target.xs], [np.asarray(u).tolist() for u in target.us]])

    def setObservation(self, t, x, u):
        self.calls.append(["setObservation", float(t), np.asarray(x).tolist(), np.asarray(u).tolist()])

    def advanceMpc(self):
        self.solves += 1
        self.calls.append(["advanceMpc"])

    def evaluateMpcSolution(self, t, x, x_opt, u_opt):
        self.calls.append(["evaluateMpcSolution", float(t), np.asarray(x).tolist()])
        x_opt[:] = 0.5 * np.asarray(x) + self.solves
        u_opt[:] = -self.solves - t

    def getMpcSolution(self, ts, xs, us):
        self.calls.append(["getMpcSolution"])
        for k in range(3):
            ts.push_back(0.1 * k); xs.push_back(np.full(self.nx, float(k))); us.push_back(np.full(self.nu, -float(k)))
