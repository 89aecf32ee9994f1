#!/usr/bin/env python3
"""Benchmark of the hot path: batched MPC solves of the headline configuration
(BASELINE.json configs[1]: Thing mobile manipulator + 1 object, horizon 20, 1024 random start states per GPU).

A "step" is one MPC solve (advanceMpc with sqp_iteration = 1, controller.yaml:56) of every instance of the
batch from its start state with the DefaultInitializer guess: linearise 21 knots -> structured IPM/Riccati
QP -> filter line search.  Inputs (start states, targets, body parameters) are resident in HBM before the
timed region; the timed region contains only device work (+ the RCCL all-gather of solved trajectories
when --gpus > 1).  One JSON line is printed by rank 0 (contract in the task description).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# algorithmic HBM bytes of the linearisation per knot and algorithmic flops of one
# IPM iteration of the QP per instance at the headline configuration H.
PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PEAK_FP64_TFLOPS = 78.6    # public MI355X fp64 vector = matrix peak (not in the local guide)


def bytes_per_knot(P):
    nx, nu, nq, ne = P.nx, P.nu, P.nq, 6 * P.nb
    return 8 * ((nx + nu) + ne * (1 + nx) + (nq + nq * (nq + 1) // 2 + 1))


def qp_flops_per_iter(P):
    nx, nu, N = P.nx, P.nu, P.N
    n_ineq = 5 * P.nc if P.nf == 3 else 0
    ric = N * ((7.0 / 3.0) * nx ** 3 + 4 * nx * nx * nu + 2 * nx * nu * nu + nu ** 3 / 3.0)
    bar = 2.0 * N * n_ineq * (nx + nu) ** 2
    return ric + bar


def pmc_traffic():
    """Per-launch HBM bytes of the two kernels from the committed rocprofv3 PMC passes (profiles/):
    (2 * FETCH_SIZE + WRITE_SIZE) * 1024, the gfx950 correction of MI355X_MICROARCH.md.  They are measured at
    the default workload (B = 1024); None when no profile is committed."""
    import csv

    best = None
    for f in sorted((ROOT / "profiles").glob("r*_pmc_hbm.csv")):
        best = f
    if best is None:
        return {}, None
    vals = {}
    for row in csv.reader(l for l in open(best) if not l.startswith("#")):
        if len(row) == 4 and row[0] != "kernel":
            vals[(row[0], row[1])] = float(row[3])
    out = {}
    for key, tag in (("upr_qp", "qp"), ("upr_linearize_kernel", "linearize")):
        fe = [v for (k, c), v in vals.items() if key in k and c == "FETCH_SIZE"]
        wr = [v for (k, c), v in vals.items() if key in k and c == "WRITE_SIZE"]
        if fe and wr:
            out[tag] = (2.0 * fe[0] + wr[0]) * 1024.0
    return out, best.name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU")
    ap.add_argument("--cpu-sample", type=int, default=512, help="instances timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as g

    if not g.LIB.exists():
        g.build()
    from upright_amd.engine import BatchMPC
    from upright_amd.problem import thing_problem
    from upright_amd.sampling import level_tray_states, stationary_guess, waypoints_for

    arr = json.load(open(ROOT / "tests" / "golden" / "arrangements.json"))["pink_bottle"]
    # nx 27, nu 21, 6 equality + 20 friction rows per knot, N = 20; sqp.use_feedback_policy as in controller.yaml:60
    P = thing_problem(arr, use_feedback_policy=True)
    B = args.batch
    # shard: rank r owns instances [r*B, (r+1)*B) of the global sample (weak scaling: B per GPU)
    x0_all = level_tray_states(B * world, seed=0)
    x0 = x0_all[rank * B:(rank + 1) * B]
    way = waypoints_for(P, x0)
    mpc = BatchMPC(P, B, way_p=way)
    mpc.set_observation(0.0, x0)

    n1 = P.N + 1
    gather_x = gather_u = loc_x = loc_u = None
    if world > 1:
        loc_x = torch.empty(B * n1 * P.nx, dtype=torch.float64, device="cuda")
        loc_u = torch.empty(B * P.N * P.nu, dtype=torch.float64, device="cuda")
        gather_x = torch.empty(world * B * n1 * P.nx, dtype=torch.float64, device="cuda")
        gather_u = torch.empty(world * B * P.N * P.nu, dtype=torch.float64, device="cuda")

    def step():
        mpc.reset_async()      # cold start: DefaultInitializer guess
        mpc.advance_async()
        if world > 1:          # exchange step: all-gather of the solved trajectories
            mpc.copy_solution_device(loc_x.data_ptr(), loc_u.data_ptr())
            mpc.sync()
            dist.all_gather_into_tensor(gather_x, loc_x)
            dist.all_gather_into_tensor(gather_u, loc_u)

    for _ in range(args.warmup):
        step()
    mpc.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    mpc.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    mpc.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kt = mpc.kernel_times()
    st = mpc.stats()
    mpc.enable_timing(False)

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        total_instances = B * world
        value = total_instances * args.steps / elapsed
        iters = float(np.sum(st["qp_iters_last"]))
        knots = B * n1 * P.sqp_iters
        lin_bytes = bytes_per_knot(P) * knots
        lin_gbs = lin_bytes / (kt["linearize_ms"] * 1e-3) / 1e9 if kt["linearize_ms"] > 0 else 0.0
        qp_flops = qp_flops_per_iter(P) * iters
        qp_tflops = qp_flops / (kt["qp_ms"] * 1e-3) / 1e12 if kt["qp_ms"] > 0 else 0.0
        traffic, traffic_src = pmc_traffic() if B == 1024 else ({}, None)
        out = {
            "metric": "batched MPC solves/sec + ms/SQP-iter, Thing 1-obj horizon=20, 1/2/4/8 GPU",   # BASELINE.json:metric verbatim; value = solves/s
            "value": value,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "ms_per_sqp_iter": ms_step / P.sqp_iters,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: Thing + pink_bottle (nx 27, nu 21, 6 eq + 20 friction rows/knot), N=20, dt=0.1, "
                            f"batch={B} level-tray random start states per GPU, cold start, sqp_iteration=1, qp iter_max=30",
                "batch_per_gpu": B,
                "parallelism": f"instances sharded over {world} rank(s); all-gather of solved trajectories" if world > 1 else "single GPU",
                "qp_converged_fraction": float(np.mean(st["qp_status_last"] == 0)),
                "qp_iters_mean": float(np.mean(st["qp_iters_last"])),
            },
            # dominant kernel = the QP (IPM/Riccati) kernel: fp64 FMA bound; achieved = SURVEY.md 8(d)
            # classical Riccati flop count x IPM iterations of the launch / its HIP-event duration
            "roofline": {
                "kernel": "upr_qp_kernel",
                "bound": "mfma",
                "achieved": qp_tflops,
                "peak": PEAK_FP64_TFLOPS,
                "unit": "TFLOP/s",
                "frac": qp_tflops / PEAK_FP64_TFLOPS,
                "traffic": traffic.get("qp"),
                "traffic_source": traffic_src,
                "avg_launch_ms": kt["qp_ms"],
            },
            # the constraint / linearisation kernel the north_star asks HBM GB/s for
            "roofline_linearize": {
                "kernel": "upr_linearize_kernel",
                "bound": "hbm",
                "achieved": lin_gbs,
                "peak": PEAK_HBM_GBS,
                "unit": "GB/s",
                "frac": lin_gbs / PEAK_HBM_GBS,
                "traffic": traffic.get("linearize"),
                "algorithmic_bytes": lin_bytes,
                "avg_launch_ms": kt["linearize_ms"],
                "bytes_per_knot": bytes_per_knot(P),
            },
            "kernel_ms": {"linearize": kt["linearize_ms"], "qp": kt["qp_ms"], "linesearch": kt["linesearch_ms"], "launches": kt["launches"]},
        }
        if not args.no_cpu_baseline and world == 1:
            import copy
            from concurrent.futures import ThreadPoolExecutor

            from oracle.oracle import Oracle

            xs0, us0 = stationary_guess(x0, P.N, P.nu)

            n = min(args.cpu_sample, B)
            solvers = []
            for b in range(n):   # one controller object per instance, built outside the timed region
                Pb = copy.copy(P)
                Pb.way_p = way[b]
                solvers.append(Oracle(Pb))

            def cpu_solve(b):
                solvers[b].solve(0.0, x0[b], xs0[b], us0[b])   # ctypes call: the GIL is released while it runs

            # (i) one thread
            tc = time.perf_counter()
            n1 = 0
            while n1 < min(64, B) and time.perf_counter() - tc < 8.0:
                cpu_solve(n1)
                n1 += 1
            dt1 = time.perf_counter() - tc
            # (ii) every host core, instances in parallel (what the reference's sequential loop would become with
            # one controller per core, planning_sim_loop.py:613-655)
            cores = os.cpu_count() or 1
            tc = time.perf_counter()
            with ThreadPoolExecutor(max_workers=cores) as ex:
                list(ex.map(cpu_solve, range(n)))
            dt_cpu = time.perf_counter() - tc
            out["cpu_baseline"] = {
                "value": n / dt_cpu,
                "unit": "solves/s",
                "cores": cores,
                "kind": "port",
                "sample": f"first {n} instances of the same batch, same cold start, oracle/upright_oracle.cpp (dense-stage Riccati IPM, -O2), "
                          f"one instance per thread on {cores} threads; reference solver (OCS2 fork + HPIPM) is not available, see BASELINE.md",
                "single_thread_value": n1 / dt1,
                "single_thread_ms_per_solve": 1e3 * dt1 / n1,
            }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
