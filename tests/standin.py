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
