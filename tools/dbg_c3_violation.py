"""configs[2] in the converging regime: WHICH rows of WHICH instances carry the constraint violation the bench line's `warm` entry reports
(VERDICT r05 item 5).  After the cold solves and the settling iterations of bench.py's configs[2] workload, per instance: the SQP's own
violation norm (stats), the worst object-dynamics residual, friction row, collision row, dynamics defect and box excess along the plan."""
import sys
sys.path.insert(0, '.')
import numpy as np
import bench
from upright_amd.engine import core_friction_rows
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = bench.config3_workload(B)
P = w["P"]
mpc = bench.make_engine(w)
for _ in range(3):
    mpc.reset(); mpc.advance()
for _ in range(12):
    mpc.advance()
st = mpc.stats()
_, xs, us = mpc.solution()
N, nq, nx = P.N, P.nq, P.nx
v = st["constraint_violation"]
order = np.argsort(-v)
x = xs[:, :N].reshape(B * N, -1); u = us.reshape(B * N, -1)
lin = mpc.linearize_points(x, u, t=np.tile(P.dt * np.arange(N), B), inst=np.repeat(np.arange(B), N))
g = np.abs(lin["g"]).reshape(B, N, -1)
fr = core_friction_rows(P, u[:, nq:]).reshape(B, N, -1) if P.nf == 3 else np.zeros((B, N, 1))
ob = mpc.obstacle_rows(xs[:, 1:N].reshape(B * (N - 1), -1), jac=False).reshape(B, N - 1, -1)
h = P.dt
q, qd, qdd, jerk = xs[:, :, :nq], xs[:, :, nq:2 * nq], xs[:, :, 2 * nq:3 * nq], us[:, :, :nq]
defect = np.abs(np.concatenate([q[:, :-1] + h * qd[:, :-1] + 0.5 * h * h * qdd[:, :-1] + h ** 3 / 6 * jerk - q[:, 1:],
                                qd[:, :-1] + h * qdd[:, :-1] + 0.5 * h * h * jerk - qd[:, 1:], qdd[:, :-1] + h * jerk - qdd[:, 1:]], axis=2)).max(axis=(1, 2))
boxx = np.maximum(np.maximum(P.x_lb - xs[:, 1:, :nx], xs[:, 1:, :nx] - P.x_ub), 0).max(axis=(1, 2))
boxu = np.maximum(np.maximum(P.u_lb - us, us - P.u_ub), 0).max(axis=(1, 2))
print("B", B, "status", np.bincount(st["qp_status_last"].astype(int), minlength=3), "violation quantiles 0.5 / 0.9 / 0.99 / 1:", np.quantile(v, [0.5, 0.9, 0.99, 1.0]))
print("instances above 1e-3:", int(np.sum(v > 1e-3)), " of them with a converged last QP:", int(np.sum((v > 1e-3) & (st["qp_status_last"] == 0))))
print("%6s %10s %6s %6s | %12s %12s %12s %12s %10s %10s | %s" % ("inst", "violation", "status", "qp it", "max |g| (eq)", "min friction", "min collis.", "defect", "box x", "box u", "knot of max |g|"))
for b in order[:12]:
    print("%6d %10.3e %6d %6d | %12.3e %12.3e %12.3e %12.3e %10.2e %10.2e | %d" % (b, v[b], st["qp_status_last"][b], st["qp_iters_last"][b], g[b].max(), fr[b].min(), ob[b].min(), defect[b], boxx[b], boxu[b],
                                                                               int(np.argmax(g[b].max(axis=1)))))
big = v > 1e-3
print("over the %d instances above 1e-3: median max|g| %.3e, median min friction %.3e, median min collision %.3e" % (big.sum(), np.median(g[big].max(axis=(1, 2))), np.median(fr[big].min(axis=(1, 2))), np.median(ob[big].min(axis=(1, 2)))))
mpc.close()
