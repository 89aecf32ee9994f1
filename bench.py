#!/usr/bin/env python3
"""Benchmark of the hot path: batched MPC solves of the headline configuration
(BASELINE.json configs[1]: Thing mobile manipulator + 1 object, horizon 20, 1024 random start states per GPU).

A "step" is one MPC solve (advanceMpc with sqp_iteration = 1, controller.yaml:56) of every instance of the
batch from its start state with the DefaultInitializer guess: linearise 21 knots -> structured IPM/Riccati
QP -> filter line search.  Inputs (start states, targets, body parameters) are resident in HBM before the
timed region; the timed region contains only device work (+ the RCCL all-gather of solved trajectories
when --gpus > 1).  One JSON line is printed by rank 0 (contract in the task description).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8                       # launches 8 ranks itself (torch.distributed.run, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

The other BASELINE configurations are timed as `extra_workloads` of the same line, each with its own roofline object:
configs[2] (Thing + 3 stacked objects + static obstacles, batch 4096; N = 1 only), configs[3] (upright_robust 8-corner
arrangement with per-instance inertial parameters, 1024 per GPU: with --gpus N the N x 1024 scenarios are sharded over the
ranks and the solved trajectories all-gathered), configs[4] (thrown ball, closed loop at 100 Hz: the goal sweep is sharded
over the ranks and only the first input u_0 of every instance is gathered per tick), and the headline problem on the start
distribution of SURVEY.md section 8(d) as written (U(+-0.25) on all nine joints; N = 1 only).

    python bench.py --gpus 2 --dry-run             # the N-rank control flow over gloo with stand-in engines (no GPU, no numbers)
"""
import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

METRIC = "batched MPC solves/sec + ms/SQP-iter, Thing 1-obj horizon=20, 1/2/4/8 GPU"   # BASELINE.json:metric verbatim; value = solves/s
PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PEAK_FP64_TFLOPS = 78.6    # public MI355X fp64 vector = matrix peak (not in the local guide)
# tools/probe/mall_probe (profiles/r06_mall_probe.txt): 512 workgroups re-streaming a private 216 KB set (the QP kernel's far-array
# pattern: 110 MB live, missing the L2, fitting the 256 MiB Infinity Cache) read it at 7.06 TB/s; the same bytes streamed from DRAM
# at 3.98 TB/s -- with identical FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ_DRAM: the counters sit in front of the Infinity Cache
MALL_PROBE_GBS = 7060.9
MALL_STREAM_GBS = 3975.6


# ---- algorithmic bytes / flops ------------------------------------------------------------
def bytes_per_knot(P):
    """Non-constant HBM bytes of one shooting knot of the linearisation: (x, u) in, g and dg/dx out, end-effector
    cost / gradient / packed Gauss-Newton Hessian out, + collision rows with their d/dq."""
    nx, nu, nq, ne = P.nx, P.nu, P.nq, 6 * P.nb
    no = len(P.pair_a) + len(P.proj_sph)
    return 8 * ((nx + nu) + ne * (1 + nx) + (nq + nq * (nq + 1) // 2 + 1) + no * (1 + nq))


def qp_flops_per_iter(P):
    """Classical dense Riccati count of one IPM iteration + barrier-Hessian update of the polytopic rows."""
    nx, nu, N = P.nx, P.nu, P.N
    n_ineq = (5 * P.nc if P.nf == 3 else 0) + len(P.pair_a) + len(P.proj_sph)
    ric = N * ((7.0 / 3.0) * nx ** 3 + 4 * nx * nx * nu + 2 * nx * nu * nu + nu ** 3 / 3.0)
    bar = 2.0 * N * n_ineq * (nx + nu) ** 2
    return ric + bar


def _latest_profile(pattern):
    files = sorted((ROOT / "profiles").glob(pattern))
    return files[-1] if files else None


def _pmc_rows(patterns):
    """(kernel, counter) -> per-dispatch mean of the newest committed counter file matching one of `patterns`."""
    import csv

    for pat in patterns:
        best = _latest_profile(pat)
        if best is None:
            continue
        vals = {}
        for row in csv.reader(l for l in open(best) if not l.startswith("#")):
            if len(row) == 4 and row[0] != "kernel":
                vals[(row[0], row[1])] = float(row[3])
        yield vals, best.name


def pmc_traffic(kernel=None):
    """Per-launch HBM bytes from the newest committed rocprofv3 PMC passes (profiles/): (2 * FETCH_SIZE + WRITE_SIZE) * 1024,
    the gfx950 correction of MI355X_MICROARCH.md.  kernel None: the headline workload's QP and linearise kernels
    (r*_pmc_hbm.csv, B = 1024); a kernel name: that QP kernel out of the all-workloads pass (r*_pmc_hbm_all.csv, mean over
    the dispatches of that pass).  {} when nothing is committed."""
    pats = ("r*_pmc_hbm.csv",) if kernel is None else ("r*_pmc_hbm_all.csv", "r*_pmc_hbm.csv")
    for vals, name in _pmc_rows(pats):
        out = {}
        for key, tag in ((kernel or "upr_qp", "qp"), ("upr_linearize", "linearize")):   # (upr_linearize_kernel<...> or upr_linearize2_kernel<nq>)
            fe = [v for (k, c), v in vals.items() if key in k and c == "FETCH_SIZE"]
            wr = [v for (k, c), v in vals.items() if key in k and c == "WRITE_SIZE"]
            if fe and wr:
                out[tag] = (2.0 * fe[0] + wr[0]) * 1024.0
        if "qp" in out:
            return out, name
    return {}, None


def pmc_workload(key):
    """Counters of ONE workload from the newest committed per-workload pass (profiles/r*_pmc_workloads.csv, tools/profile_all.sh:
    `bench.py --only <key>` under rocprofv3 --pmc, separate passes for FETCH_SIZE, WRITE_SIZE and the instruction mix; per-dispatch
    means).  Returns ({"qp": {counter: mean}, "linearize": {...}, "batch": B of the pass}, file name) or ({}, None)."""
    import csv

    best = _latest_profile("r*_pmc_workloads.csv")
    if best is None:
        return {}, None
    out = {"qp": {}, "linearize": {}}
    for row in csv.reader(l for l in open(best) if not l.startswith("#")):
        if len(row) == 6 and row[0] == key:
            _, kernel, counter, _, mean, batch = row
            tag = "linearize" if "upr_linearize" in kernel else ("qp" if "upr_qp" in kernel else None)
            if tag:
                out[tag][counter] = float(mean)
                out["batch"] = int(batch)
                if tag == "linearize":
                    out["linearize_kernel"] = kernel.replace("void ", "")   # (which of the two linearisation kernels the workload ran)
    for l in open(best):
        if l.startswith("# kernel sources sha256:"):
            out["library"] = l.split(":", 1)[1].strip()   # the sources the counters' build was made of (tools/pmc_workloads.sh)
    return (out, best.name) if (out["qp"] or out["linearize"]) else ({}, None)


def library_sha():
    """sha256 over the kernel sources (csrc/*.h, csrc/*.hip, include/upright_mi.h): what a counter file names as its build."""
    import hashlib

    h = hashlib.sha256()
    import re

    rel = sorted([str(f.relative_to(ROOT)) for f in list((ROOT / "upright_amd" / "csrc").glob("*.h")) + list((ROOT / "upright_amd" / "csrc").glob("*.hip"))] + ["include/upright_mi.h"])
    for f in rel:
        # (comments and white space do not make a build: an edit of a comment must not orphan the counters)
        code = re.sub(r"/\*.*?\*/", "", (ROOT / f).read_text(), flags=re.S)
        code = re.sub(r"//[^\n]*", "", code)
        h.update(re.sub(r"\s+", " ", code).encode())
    return h.hexdigest()[:16]


def _hbm_bytes(c):
    """gfx950 correction of MI355X_MICROARCH.md: (2 FETCH_SIZE + WRITE_SIZE) KB."""
    return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 if ("FETCH_SIZE" in c and "WRITE_SIZE" in c) else None


def _issued_flops(c):
    if "SQ_INSTS_VALU_FMA_F64" not in c:
        return None
    return 64.0 * (2.0 * c["SQ_INSTS_VALU_FMA_F64"] + c.get("SQ_INSTS_VALU_MUL_F64", 0.0) + c.get("SQ_INSTS_VALU_ADD_F64", 0.0)) + 2048.0 * c.get("SQ_INSTS_VALU_MFMA_F64", 0.0)


def pmc_issued_flops(kernel=None):
    """fp64 flops the QP kernel ISSUED per launch according to the newest committed instruction-mix pass
    (profiles/r*_pmc_mfma.csv, or r*_pmc_mfma_all.csv for a named kernel of the other workloads; wave instructions x 64
    lanes: FMA 2 flops, MUL / ADD 1, one v_mfma_f64_16x16x4 = 2048)."""
    pats = ("r*_pmc_mfma.csv",) if kernel is None else ("r*_pmc_mfma_all.csv", "r*_pmc_mfma.csv")
    for vals, name in _pmc_rows(pats):
        c = {cn: v for (k, cn), v in vals.items() if (kernel or "upr_qp") in k}
        if "SQ_INSTS_VALU_FMA_F64" not in c:
            continue
        flops = 64.0 * (2.0 * c["SQ_INSTS_VALU_FMA_F64"] + c.get("SQ_INSTS_VALU_MUL_F64", 0.0) + c.get("SQ_INSTS_VALU_ADD_F64", 0.0)) \
            + 2048.0 * c.get("SQ_INSTS_VALU_MFMA_F64", 0.0)
        return flops, name
    return None, None


# ---- workloads ----------------------------------------------------------------------------------------------------
def _arrangements():
    return json.load(open(ROOT / "tests" / "golden" / "arrangements.json"))


def headline_workload(B, rank=0, world=1):
    """configs[1]: nx 27, nu 21, 6 equality + 20 friction rows per knot, N = 20; sqp.use_feedback_policy as in
    controller.yaml:60.  Rank r owns instances [r B, (r+1) B) of the global sample (weak scaling)."""
    from upright_amd.distributed import shard_range
    from upright_amd.problem import thing_problem
    from upright_amd.sampling import level_tray_states, waypoints_for

    P = thing_problem(_arrangements()["pink_bottle"], use_feedback_policy=True)
    lo, hi = shard_range(B * world, rank, world)
    x0 = level_tray_states(B * world, seed=0)[lo:hi]
    return dict(P=P, x0=x0, way=waypoints_for(P, x0), body_params=None,
                name="configs[1]: Thing + pink_bottle (nx 27, nu 21, 6 eq + 20 friction rows/knot), N=20, dt=0.1, "
                     f"batch={B} level-tray random start states per GPU, cold start, sqp_iteration=1, qp iter_max=30")


def headline_r03_geometry_workload(B):
    """The headline workload as rounds 1 - 3 measured it: the arm mounted at yaw 0 (round 4 fixed the mount at -pi/2 from the
    reference's own scene, upright_amd/robots.py; the level-tray starts and the target offset are the same).  Kept so that the
    kernels can be compared across rounds on equal work: the interior-point iteration counts of the two batches differ
    (10.2 against 10.5 on average, longest 13 against 16), and a launch lasts as long as its slowest pair of instances."""
    from upright_amd.problem import thing_problem
    from upright_amd.sampling import level_tray_states, waypoints_for

    P = thing_problem(_arrangements()["pink_bottle"], use_feedback_policy=True, mount_yaw=0.0)
    x0 = level_tray_states(B, seed=0)
    return dict(P=P, x0=x0, way=waypoints_for(P, x0), body_params=None, key="headline_r03",
                name=f"configs[1] with the arm mount of rounds 1 - 3 (yaw 0; same starts and target offset): Thing + pink_bottle, N=20, batch={B}, cold start, sqp_iteration=1")


def config3_workload(B):
    """configs[2]: box_arch (3 bodies, 16 contact points, arrangements.yaml:8-51) + the 20 sphere pairs of
    obstacles/simple.yaml:11-41, waypoint _point3 [0, -2, 0.25], seed 1."""
    from upright_amd import robots
    from upright_amd.problem import THING_HOME, thing_problem
    from upright_amd.sampling import waypoints_for

    P = thing_problem(_arrangements()["box_arch"])
    for k, v in robots.collision_model(P.chain, robots.SIMPLE_COLLISION_PAIRS).items():
        setattr(P, k, v)
    rng = np.random.default_rng(1)
    x0 = np.tile(np.concatenate([THING_HOME, np.zeros(18)]), (B, 1))
    x0[:, :2] += rng.uniform(-0.08, 0.08, (B, 2))   # around the stock home pose (round 4: the arm mount that clears every pair there, upright_amd/robots.py)
    return dict(P=P, x0=x0, way=waypoints_for(P, x0, offset=(0.0, -2.0, 0.25)), body_params=None, key="config3",
                name=f"configs[2]: Thing + box_arch (3 bodies, 16 contacts: nx 27, nu 57, 18 eq + 80 friction + 20 collision rows/knot), "
                     f"N=20, batch={B}, cold start, sqp_iteration=1")


def config4_workload(B, rank=0, world=1):
    """configs[3]: the upright_robust 8-corner arrangement (planning_sim_loop.py:454-534), frictionless, per-instance
    inertial parameters (CoM uniform in the CoM box, inertia scaled by {1, 0.5, 0.1}, :559), HPIPM slacks on the state
    boxes and the general constraints, init_sqp_iteration 3 (upright_robust/config/demos/_base.yaml:62-75), waypoint
    [-2, 1, 0], seed 2.  Rank r owns scenarios [r B, (r+1) B) of the global sample of world x B (8 x 1024 = BASELINE's 8192;
    planning_sim_loop.py:613-655 runs them one after the other)."""
    from upright_amd.distributed import shard_range
    from upright_amd.problem import THING_HOME, thing_problem
    from upright_amd.sampling import waypoints_for

    P = thing_problem(_arrangements()["robust_8corner"], nf=1, force_weight=0.0, sqp_iters=3)
    P.slacks = dict(state_box=True, input_box=False, poly_ineq=True)
    rng = np.random.default_rng(2)
    Bg = B * world
    bp = np.zeros((Bg, 8, 10))
    for b in range(Bg):
        sc = (1.0, 0.5, 0.1)[b % 3]
        for i in range(8):
            com = rng.uniform([-0.06, -0.06, -0.15], [0.06, 0.06, 0.15])
            bp[b, i] = [1.0, *com, sc * 0.009375, 0, 0, sc * 0.009375, 0, sc * 0.00375]
    x0 = np.tile(np.concatenate([THING_HOME, np.zeros(18)]), (Bg, 1))
    x0[:, :2] += rng.uniform(-0.25, 0.25, (Bg, 2))
    lo, hi = shard_range(Bg, rank, world)
    x0, bp = x0[lo:hi], bp[lo:hi]
    return dict(P=P, x0=x0, way=waypoints_for(P, x0, offset=(-2.0, 1.0, 0.0)), body_params=bp, key="config4",
                name=f"configs[3]: upright_robust 8-corner arrangement (8 bodies, 32 frictionless contacts: nx 27, nu 41, 48 soft eq rows/knot), "
                     f"per-instance inertial parameters, N=20, batch={B} per GPU, cold start, init_sqp_iteration=3")


def config5_workload(B, rank=0, world=1, slacks=False):
    """configs[4]: Thing + pink_bottle with a thrown ball (dynamic obstacle, obstacles/dynamic.yaml:5-17; rows as in
    ral23/experiments/projectile/_base.yaml:81-87: two self-collision pairs, wrist-vs-ground, forearm-vs-ball and the
    projectile-path row on the tray's link), closed loop at 100 Hz; goal sweep: goals on a 1.2 m circle around the
    start pose of the tray, the ball crosses each instance's straight tray path one second from the start."""
    from upright_amd import robots
    from upright_amd.problem import THING_HOME, thing_problem

    P = thing_problem(_arrangements()["pink_bottle"], use_feedback_policy=True)
    pairs = [("wrist1_collision_link_0", "shoulder_collision_link_0"), ("wrist3_collision_link_0", "ground"),
             ("forearm_collision_sphere_link2_0", "projectile1")]
    for k, v in robots.collision_model(P.chain, pairs, dynamic={"projectile1": 0.2}).items():
        setattr(P, k, v)
    P.n_dyn = 1
    robots.add_projectile_rows(P, ["balanced_object_collision_link"], [0.35], 0.2)
    if slacks:   # sqp.hpipm.slacks.enabled with the reference's defaults (controller.yaml:68-72): boxes and polytopic rows softened
        P.slacks = dict(state_box=True, input_box=True, poly_ineq=True)
    x0 = np.tile(np.concatenate([THING_HOME, np.zeros(18)]), (B, 1))
    p, _ = P.chain.forward(THING_HOME)
    ang = 2.0 * np.pi * (rank * B + np.arange(B)) / (B * world)   # (rank r owns goals [r B, (r+1) B) of the sweep of world x B)
    goal = np.stack([1.2 * np.cos(ang), 1.2 * np.sin(ang), np.zeros(B)], axis=1)
    T = 1.0
    a0 = np.array([0.0, 0.0, -9.81])
    dyn = np.zeros((B, 9))
    for b in range(B):
        # thrown across the tray's path (perpendicular to it in the plane), through the point half way to the goal
        perp = np.array([-np.sin(ang[b]), np.cos(ang[b]), 0.0])
        v0 = 2.5 * perp + np.array([0.0, 0.0, 0.5 * 9.81 * T])
        cross = p + 0.5 * goal[b] + np.array([0.0, 0.0, 0.25])
        dyn[b] = np.concatenate([cross - v0 * T - 0.5 * a0 * T * T, v0, a0])
    return dict(P=P, x0=np.concatenate([x0, dyn], axis=1), way=(p + goal)[:, None, :], body_params=None, key="config5s" if slacks else "config5",
                name=f"configs[4]: Thing + pink_bottle + thrown ball (5 collision / projectile rows per knot), N=20, closed loop at 100 Hz "
                     f"(one warm-started SQP iteration per tick, linear feedback policy at the observed state), goal sweep of {B} goals per GPU"
                     + (", sqp.hpipm.slacks.enabled (state_box, input_box, poly_ineq: the reference's remedy for rows one step cannot restore)" if slacks else ""))


def contract_workload(B):
    """The headline problem on the start distribution of SURVEY.md section 8(d) AS WRITTEN: x_home + U(-0.25, 0.25) on all nine
    joints, U(-0.2, 0.2) on the velocities, zero accelerations, seed 0.  Most of these starts tilt the tray beyond the
    friction cone at the fixed first knot (upright_amd/sampling.py): their hard-constrained QPs are infeasible, for the
    reference's formulation as for this one, and are reported per instance."""
    from upright_amd.problem import thing_problem
    from upright_amd.sampling import contract_states, waypoints_for

    P = thing_problem(_arrangements()["pink_bottle"], use_feedback_policy=True)
    x0 = contract_states(B, seed=0)
    return dict(P=P, x0=x0, way=waypoints_for(P, x0), body_params=None, key="contract",
                name=f"configs[1] on SURVEY 8(d)'s start distribution as written (U(+-0.25) on all nine joints, U(+-0.2) velocities, seed 0), "
                     f"batch={B}, cold start, sqp_iteration=1: most starts are infeasible at the fixed first knot (tray tilted beyond the friction cone)")


def time_closed_loop(w, ticks, dist=None, device="cuda", engine=None, force_exchange=False):
    """configs[4]: `ticks` control periods of 10 ms for the whole batch: observation in (host -> device), one warm-started
    SQP iteration, policy out (device -> host), exact triple-integrator plant and ballistic ball on the host.  The
    plant's states come from outside the engine every tick, so this rate includes both PCIe hops by construction.  One
    `upr_batch_tick` per period (= setObservation + advanceMpc + evaluateMpcSolution of manager.py:156-176, one synchronisation)."""
    mpc = engine if engine is not None else make_engine(w)
    P, B = w["P"], mpc.B
    world = dist.get_world_size() if dist is not None else 1
    exchange = world > 1 or (force_exchange and dist is not None)     # (force_exchange: the one-GPU test of the nccl branch)
    mpc.set_projectile_flag(1.0)
    x, t, dt = w["x0"].copy(), 0.0, 0.01
    failed = 0
    broke = [0]
    lat = []
    lat_x = []
    u0_all = [None]
    stq_ = [None]
    u_dev = u_out = ext = None
    if exchange:
        # exchange buffers of the closed loop: u_0 stays on the device (engine buffer -> u_dev on the engine's stream -> RCCL)
        import torch

        from upright_amd.distributed import all_gather_first_inputs

        u_dev = torch.empty((B, P.nu), dtype=torch.float64, device=device)
        u_out = torch.empty((world * B, P.nu), dtype=torch.float64, device=device)
        if device != "cpu" and hasattr(mpc, "stream_ptr") and mpc.stream_ptr():
            ext = torch.cuda.ExternalStream(mpc.stream_ptr())

    def tick(x, t):
        tc = time.perf_counter()
        _, u, st_ = mpc.tick(t, x, want_stats=True)    # observation in, one SQP iteration, policy at the observation + statistics out: one call
        stq_[0] = st_["qp_status_last"]
        lat.append(time.perf_counter() - tc)
        if exchange:   # exchange step of the closed loop: only u_0 of every instance, [B, nu] per rank,
            # gathered from the engine's DEVICE buffer (no second trip over PCIe); the tick's latency with the exchange is
            # stamped only when the collective has completed
            mpc.copy_policy_device(u_dev.data_ptr())
            if ext is not None:
                torch.cuda.current_stream().wait_stream(ext)
            else:
                mpc.sync()
            u0_all[0], hd = all_gather_first_inputs(u_dev, out=u_out, async_op=True)
            hd.wait()
            if device != "cpu":
                torch.cuda.current_stream().synchronize()
            if ext is not None:
                ext.wait_stream(torch.cuda.current_stream())      # u_dev is free for the next tick's copy
            lat_x.append(time.perf_counter() - tc)
        j = u[:, :9]
        q, v, a = x[:, :9], x[:, 9:18], x[:, 18:27]
        ro, vo, ao = x[:, 27:30], x[:, 30:33], x[:, 33:36]
        return np.concatenate([q + dt * v + dt ** 2 / 2 * a + dt ** 3 / 6 * j, v + dt * a + dt ** 2 / 2 * j, a + dt * j,
                               ro + dt * vo + 0.5 * dt * dt * ao, vo + dt * ao, ao], axis=1)

    def run_pass(events):
        # one run of the closed loop from the workload's start states.  events: per-kernel HIP events on (kernel_ms of the line);
        # off, the engine replays each control period as one captured HIP graph (upr_batch_tick)
        nf = nb = 0
        mpc.reset_async()
        x, t = w["x0"].copy(), 0.0
        x = tick(x, t); t += dt          # first solve: cold start + allocation effects, untimed
        lat.clear(); lat_x.clear()
        mpc.enable_timing(events)
        t0 = time.perf_counter()
        for _ in range(ticks):
            x = tick(x, t); t += dt
            stq = stq_[0]
            nf += int(np.sum(stq != 0)); nb += int(np.sum(stq == 2))
        return time.perf_counter() - t0, nf, nb, x

    # the kernels' own times come from a SECOND, identical run with events (they cost ~20 us per period and switch the graph
    # replay off); the rate, the latencies and the status counts are the first run's
    _, _, _, _ = run_pass(True)
    kt = mpc.kernel_times()
    elapsed, failed, nbroke, x = run_pass(False)
    broke[0] = nbroke
    graph_replays = mpc.tick_graph_replays() if hasattr(mpc, "tick_graph_replays") else 0
    if world > 1:
        import torch

        tt = torch.tensor([elapsed, float(failed), float(broke[0])], dtype=torch.float64, device=device)
        dist.all_reduce(tt[:1], op=dist.ReduceOp.MAX)      # the slowest rank's clock
        dist.all_reduce(tt[1:], op=dist.ReduceOp.SUM)
        elapsed, failed, broke[0] = float(tt[0].item()), int(tt[1].item()), int(tt[2].item())
        assert u0_all[0].shape == (world * B, P.nu)
    roof, lin = roofline_objects(P, B, kt, mpc.stats(), 1, headline=False, key=w.get("key"))   # (kernel times: means over the ticks; IPM iterations: the last tick's)
    goal_err = np.linalg.norm(np.array([P.chain.forward(x[b, :9])[0] for b in range(0, B, max(1, B // 64))])
                              - w["way"][::max(1, B // 64), 0], axis=1)
    out = {
        "workload": w["name"], "value": B * world * ticks / elapsed, "unit": "solves/s", "n_gpus": world, "ms_per_tick": 1e3 * elapsed / ticks,
        "ms_per_tick_p99_engine": 1e3 * float(np.quantile(lat, 0.99)),
        "ms_per_tick_p99_with_exchange": 1e3 * float(np.quantile(lat_x, 0.99)) if lat_x else None,
        "control_period_ms": 10.0, "ticks": ticks, "periods_replayed_as_hip_graph": graph_replays,
        "kernel_ms_source": "a second, identical run of the loop with per-kernel HIP events (the timed run has none)",
        "real_time_factor": 0.01 * ticks / elapsed,
        "exchange": "all-gather of u_0 per tick" if exchange else None,
        "qp_not_converged_fraction": failed / (B * world * ticks),
        "qp_factorisation_broke_down_fraction": broke[0] / (B * world * ticks),   # (status 2: no step, no feedback policy for that tick)
        "tray_to_goal_m_after_run": {"mean": float(goal_err.mean()), "max": float(goal_err.max())},
        "finite": bool(np.all(np.isfinite(x))),
        "roofline": roof, "roofline_linearize": lin,
        "kernel_ms": {"linearize": kt["linearize_ms"], "qp": kt["qp_ms"], "linesearch": kt["linesearch_ms"], "launches": kt["launches"]},
    }
    if u0_all[0] is not None:
        out["u0_gathered"] = u0_all[0]        # (for the tests; dropped before the line is printed)
        out["gathered_rows"] = int(u0_all[0].shape[0])   # first inputs of every instance of every rank, on this rank
    mpc.close()
    return out


def make_engine(w, device_index=None):
    """The engine of one rank, on HIP device `device_index` (the launcher's LOCAL_RANK; None: the current device)."""
    from upright_amd.engine import BatchMPC

    mpc = BatchMPC(w["P"], len(w["x0"]), way_p=w["way"], body_params=w["body_params"], device=device_index)
    mpc.set_observation(0.0, w["x0"])
    return mpc


# ---- the timed loop of one rank -----------------------------------------------------------------------------------
def rank_main(args, mpc, P, dist=None, device="cuda", sync_device=None, force_exchange=False, events=3):
    """W untimed + K timed steps on this rank.  `mpc` is the engine (or a stand-in with the same methods: the gloo test
    drives this function with fake solutions); `dist` a torch.distributed module with an initialised group or None.
    Returns (max-over-ranks seconds of the timed region, gathered trajectories of the last step or None).
    Exchange step (world > 1; force_exchange: also at world size 1, for the one-GPU test of the nccl branch): the copy-out of
    step i runs on the engine's stream, the collective's stream waits for it through an event -- no host synchronisation --
    and the collective runs while the engine is already in step i + 1.  Two send-buffer pairs are used in turn; before pair
    i & 1 is overwritten the handles of the collective that last read it are waited for and the engine's stream is ordered
    behind them."""
    import torch

    from upright_amd.distributed import all_gather_solutions

    world = dist.get_world_size() if dist is not None else 1
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s)")
    exchange = world > 1 or (force_exchange and dist is not None)
    B, n1 = mpc.B, P.N + 1
    loc = None
    counts = [B] * world
    ext = None
    if exchange:
        loc = [(torch.empty((B, n1, P.nx), dtype=torch.float64, device=device),
                torch.empty((B, P.N, P.nu), dtype=torch.float64, device=device)) for _ in range(2)]
        if device != "cpu" and hasattr(mpc, "stream_ptr") and mpc.stream_ptr():
            ext = torch.cuda.ExternalStream(mpc.stream_ptr())
    gathered = [None]
    pending = [None, None]
    turn = [0]

    def retire(i):
        if pending[i] is not None:
            for hd in pending[i]:
                hd.wait()                    # nccl: torch's current stream waits for the collective; gloo: the host does
            pending[i] = None
            if ext is not None:
                ext.wait_stream(torch.cuda.current_stream())

    def step():
        mpc.reset_async()      # cold start: DefaultInitializer guess
        mpc.advance_async()
        if exchange:           # exchange step: all-gather of the solved trajectories
            i = turn[0] & 1
            turn[0] += 1
            retire(i)
            loc_x, loc_u = loc[i]
            mpc.copy_solution_device(loc_x.data_ptr(), loc_u.data_ptr())
            if ext is not None:
                torch.cuda.current_stream().wait_stream(ext)
            else:
                mpc.sync()
            gx, gu, _, hs = all_gather_solutions(loc_x, loc_u, counts=counts, async_op=True)
            pending[i] = hs
            gathered[0] = (gx, gu)

    def fence():
        retire(0); retire(1)
        mpc.sync()
        if sync_device is not None:
            sync_device()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    # events = 3 (the headline): around every fourth launch of the QP kernel (the dominant kernel: roofline.achieved is its
    # average duration over THIS region); the other two kernels are timed by aux_kernel_times() behind the region.  A recorded
    # event holds the stream for 3 - 4 us (tools/exp_timing_modes.py: 1.8787 ms per solve without events, 1.8806 in this mode,
    # 1.8848 with events around every QP launch, 1.9020 around every kernel) -- six per step were 1.2 % of it for durations the
    # roofline of the dominant kernel does not need.  events = 1 (the other workloads): around every kernel
    mpc.enable_timing(events)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed, gathered[0]


def aux_kernel_times(mpc, kt_qp, steps=10):
    """Durations of the linearisation and line-search kernels: `steps` more cold solves of the same batch with events around
    every kernel, BEHIND the timed region (whose events sit around the QP kernel only).  Returns the merged table: QP from the
    timed region, the other two from this pass."""
    mpc.enable_timing(1)
    for _ in range(steps):
        mpc.reset_async(); mpc.advance_async()
    mpc.sync()
    ka = mpc.kernel_times()
    kt = dict(kt_qp)
    kt["linearize_ms"], kt["linesearch_ms"] = ka["linearize_ms"], ka["linesearch_ms"]
    kt["launches"] = [ka["launches"][0], kt_qp["launches"][1], ka["launches"][2]]
    kt["aux_qp_ms"] = ka["qp_ms"]
    return kt


def roofline_objects(P, B, kt, st, sqp_iters, headline, key=None):
    """`roofline` (QP kernel: the dominant one) and `roofline_linearize` objects of one workload.  key: the workload's name in the
    per-workload counter file (pmc_workload); counters are used only when the pass ran the same batch size."""
    knots = B * (P.N + 1)                                  # one linearise launch covers every knot of the batch once
    lin_bytes = bytes_per_knot(P) * knots
    lin_gbs = lin_bytes / (kt["linearize_ms"] * 1e-3) / 1e9 if kt["linearize_ms"] > 0 else 0.0
    # IPM iterations of the LAST QP launch of the step, per instance (the launches of a step run similar counts)
    qp_flops = qp_flops_per_iter(P) * float(np.sum(st["qp_iters_last"]))
    qp_tflops = qp_flops / (kt["qp_ms"] * 1e-3) / 1e12 if kt["qp_ms"] > 0 else 0.0
    kname = kt.get("qp_kernel", "upr_qp3_kernel")
    # counters: the headline workload's own pass; for the other workloads the all-workloads pass, looked up by kernel name
    # (template arguments as rocprofv3 prints them)
    kkey = None if headline else kname.split("upr_qp3_kernel<")[-1].rstrip(">").strip() if "upr_qp3_kernel<" in kname else kname
    # (the name-keyed files of rounds 2 - 3 only when the workload has no key of its own: a kernel name is shared by workloads
    #  whose launches differ -- the per-workload file of round 4 is the one that says which launch the counters belong to)
    legacy = key is None and (not headline or B == 1024)
    traffic, traffic_src = pmc_traffic(kkey) if legacy else ({}, None)
    issued, issued_src = pmc_issued_flops(kkey) if legacy else (None, None)
    wl, wl_src = pmc_workload(key or ("headline" if headline else ""))
    if wl and wl.get("batch") == B:          # round 4: counters of this very workload at this very batch size
        traffic, traffic_src = {"qp": _hbm_bytes(wl["qp"]), "linearize": _hbm_bytes(wl["linearize"])}, wl_src
        issued, issued_src = _issued_flops(wl["qp"]), wl_src
    else:
        wl = {}
    frac = qp_tflops / PEAK_FP64_TFLOPS
    frac_hbm = (traffic["qp"] / (kt["qp_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS) if (traffic.get("qp") and kt["qp_ms"] > 0) else None
    roof = {
        "kernel": kname,
        # `achieved` prices the SURVEY.md 8(d) classical dense Riccati count x the IPM iterations of the launch against its
        # HIP-event duration: an algorithmic-model rate (the contract's definition), not executed flops -- those are
        # `achieved_issued`.  `bound` names the roofline the kernel sits closer to (fp64 "mfma" = vector = matrix peak, or
        # "hbm" by the counter traffic); it reaches neither: `bound_detail`.
        # neither roofline is approached (executed flops <= 14 % of the fp64 peak, counter traffic mostly L2 /
        # Infinity-Cache hits on dependent chains) -- the kernel is LATENCY bound and the label says so; `nearest_roofline`
        # names the roofline it sits closer to, judged by the EXECUTED flops when the instruction-mix counters exist
        # (the model count overstates them) and by the model count otherwise.
        "bound": "latency",
        "nearest_roofline": None,
        "bound_detail": "latency / issue bound: dependent fp64 chains of the Riccati recursion (one or two waves per instance carry "
                        "the serial sweeps), far from both the fp64 and the HBM roofline",
        "achieved": qp_tflops,
        "peak": PEAK_FP64_TFLOPS,
        "unit": "TFLOP/s",
        "frac": frac,
        "traffic": traffic.get("qp"),
        "traffic_source": traffic_src,
        "frac_hbm": frac_hbm,
        "avg_launch_ms": kt["qp_ms"],
        "algorithmic_flops_per_launch": qp_flops,
    }
    # What the counter traffic is.  FETCH_SIZE / WRITE_SIZE count requests at the L2's fabric side: Infinity-Cache
    # hits included, and no gfx950 counter splits them from DRAM (rocprofv3 --list-avail: TCC_EA0_RDREQ_DRAM = "destined for DRAM",
    # equal to RDREQ for an Infinity-Cache-resident set too -- profiles/r06_mall_probe.txt).  The split is therefore a MODEL, with
    # the probe as its evidence: the workgroups resident at a time (one or two per CU) re-stream a private workspace each; their
    # live set (resident workgroups x workspace) fits the 256 MiB cache, so after the first touch the far traffic is cache
    # traffic, and DRAM sees every byte once: the linearisation records and trajectories read, the workspace written back.
    lin_stride = 6 * P.nb * (1 + P.nx) + 1 + P.nq + P.nq * (P.nq + 1) // 2 + (len(P.pair_a) + len(P.proj_sph)) * (1 + P.nq)
    ws_d = kt.get("ws_doubles") or 0
    per_inst = 8.0 * ((P.N + 1) * lin_stride + (P.N + 1) * P.nx + P.N * P.nu + ws_d)
    resident = min(B, 256 * (2 if ("upr_qp3_cfg<9, 1," in kname or "upr_qp3_cfg<6, 1," in kname) else 1))
    roof["traffic_is"] = "bytes at the L2's fabric side (2 x FETCH_SIZE + WRITE_SIZE): Infinity-Cache hits are counted, no gfx950 counter splits them off"
    roof["traffic_dram"] = B * per_inst
    roof["traffic_dram_basis"] = ("model: every instance's records, trajectories and workspace (%d doubles) cross the DRAM interface once; the live set of the "
                                  "%d resident workgroups is %.0f MB < 256 MiB Infinity Cache, which serves the re-streamed far arrays "
                                  "(profiles/r06_mall_probe.txt: that pattern reads at %.1f TB/s from the cache, %.1f TB/s from DRAM, same counters)"
                                  % (ws_d, resident, resident * 8.0 * ws_d / 1e6, MALL_PROBE_GBS / 1e3, MALL_STREAM_GBS / 1e3))
    if traffic.get("qp") and kt["qp_ms"] > 0:
        roof["frac_infinity_cache"] = traffic["qp"] / (kt["qp_ms"] * 1e-3) / 1e9 / MALL_PROBE_GBS   # of the probe's measured rate for this access pattern
        roof["frac_hbm_dram_model"] = roof["traffic_dram"] / (kt["qp_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS
    if wl.get("library"):
        # which build the counters belong to, and whether it is the one being timed
        roof["counters_of_build"] = "kernel sources sha256 " + wl["library"]
        roof["counters_match_this_build"] = (wl["library"] == library_sha())
    if issued is not None and kt["qp_ms"] > 0:
        # flops the kernel actually issued (committed instruction-mix counters of the same kernel; for workloads whose
        # launches differ in IPM iterations, e.g. configs[2] cold / warm, the mean over the dispatches of the counter pass):
        # the block structure removes about half of the classical count
        roof["issued_flops_per_launch"] = issued
        roof["achieved_issued"] = issued / (kt["qp_ms"] * 1e-3) / 1e12
        roof["frac_issued"] = roof["achieved_issued"] / PEAK_FP64_TFLOPS
        roof["issued_source"] = issued_src
    # where the SURVEY 8(d) dense count is more than 3x what the structured kernel executes (configs[2]: the barrier
    # term of 80 friction + 20 collision rows over (nx + nu)^2 = 84^2 entries -- a structure-exploiting kernel never forms it), a
    # fraction of the fp64 peak priced on that count says nothing about the kernel: `frac` is then the ISSUED fraction and the
    # model-count figure is kept beside it as `frac_model`
    roof["frac_model"] = frac
    roof["frac_basis"] = "SURVEY 8(d) algorithmic count"
    if roof.get("issued_flops_per_launch") and qp_flops / roof["issued_flops_per_launch"] > 3.0:
        roof["frac"] = roof["frac_issued"]
        roof["frac_basis"] = "issued flops (the SURVEY 8(d) dense count is %.1fx what the structured kernel executes)" % (qp_flops / roof["issued_flops_per_launch"])
    f_exec = roof.get("frac_issued", frac)
    roof["nearest_roofline"] = "hbm" if (frac_hbm is not None and frac_hbm > f_exec) else "mfma"
    if (roof.get("frac_infinity_cache") or 0.0) >= 0.75 and (roof.get("frac_infinity_cache") or 0.0) > f_exec:
        # three quarters and more of what the Infinity Cache delivered to the probe's copy of this access pattern: bandwidth of the
        # cache, not of the HBM (whose share is `frac_hbm_dram_model`), and not latency.  (The hard-row kernels sit at ~0.5 of it and
        # stay "latency": a fully L2-resident batch is only 10 % faster per round, and far-memory prefetches measured +- 0.5 %.)
        roof["bound"] = "infinity-cache"
        roof["bound_detail"] = ("fabric traffic at %.0f %% of the rate the Infinity Cache sustained for the same re-streaming pattern (tools/probe/mall_probe); "
                                "DRAM itself carries the modelled %.2f GB per launch" % (100 * roof["frac_infinity_cache"], roof["traffic_dram"] / 1e9))
    elif f_exec >= 0.5:
        roof["bound"] = "mfma"
    lin = {
        "kernel": (wl.get("linearize_kernel") if wl else None) or "upr_linearize_kernel",
        "bound": "hbm",
        "achieved": lin_gbs,
        "peak": PEAK_HBM_GBS,
        "unit": "GB/s",
        "frac": lin_gbs / PEAK_HBM_GBS,
        "traffic": traffic.get("linearize") if (headline or wl) else None,
        "traffic_source": traffic_src if (headline or wl) else None,
        "algorithmic_bytes": lin_bytes,
        "avg_launch_ms": kt["linearize_ms"],
        "bytes_per_knot": bytes_per_knot(P),
    }
    lc = wl.get("linearize", {}) if wl else {}
    if "SQ_BUSY_CYCLES" in lc:
        # MFMA utilisation of the Gauss-Newton Hessian assembly (north_star): matrix-pipe busy cycles over the SQ-busy cycles of the
        # launch (both summed over the chip), and the v_mfma_f64_16x16x4_f64 instructions behind it (one per knot and error-row block)
        lin["mfma_util"] = lc.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / lc["SQ_BUSY_CYCLES"] if lc["SQ_BUSY_CYCLES"] > 0 else None
        lin["mfma_instructions"] = lc.get("SQ_INSTS_VALU_MFMA_F64")
        lin["valu_instructions"] = lc.get("SQ_INSTS_VALU")
    return roof, lin


def time_extra(w, steps, warmup, warm=None, dist=None, device="cuda", engine=None):
    """One more BASELINE configuration on this GPU: solves/s + its own roofline objects.  warm = (n_settle, n_timed):
    after the cold-start steps, n_settle further SQP iterations from the plan found so far (no reset) and n_timed timed
    ones -- the closed-loop regime, in which the sub-problems are feasible and converge."""
    world = dist.get_world_size() if dist is not None else 1
    args = argparse.Namespace(gpus=world, steps=steps, warmup=warmup)
    mpc = engine if engine is not None else make_engine(w)
    import torch

    # (the engine's own stream sync closes the timed region; torch's is added when torch owns a context in this process)
    elapsed, gathered = rank_main(args, mpc, w["P"], dist=dist, device=device,
                                  sync_device=torch.cuda.synchronize if (device != "cpu" and torch.cuda.is_initialized()) else None, events=1)
    kt, st = mpc.kernel_times(), mpc.stats()
    B = mpc.B
    if world > 1:
        assert gathered is not None and gathered[0].shape[0] == world * B
    cold_key = (w.get("key") + "_cold") if (warm is not None and w.get("key")) else w.get("key")   # (the counter pass keeps cold and warm launches apart)
    roof, lin = roofline_objects(w["P"], B, kt, st, w["P"].sqp_iters, headline=False, key=cold_key)
    out = {
        "workload": w["name"], "value": B * world * steps / elapsed, "unit": "solves/s", "n_gpus": world,
        "exchange": "all-gather of solved trajectories" if world > 1 else None, "ms_per_step": 1e3 * elapsed / steps,
        "ms_per_sqp_iter": 1e3 * elapsed / steps / max(1, kt["launches"][1] // steps), "steps": steps, "warmup": warmup,
        "qp_converged_fraction": float(np.mean(st["qp_status_last"] == 0)), "qp_iters_mean": float(np.mean(st["qp_iters_last"])),
        "roofline": roof, "roofline_linearize": lin,
        "kernel_ms": {"linearize": kt["linearize_ms"], "qp": kt["qp_ms"], "linesearch": kt["linesearch_ms"], "launches": kt["launches"]},
    }
    if gathered is not None:
        out["gathered"] = gathered            # (for the tests; dropped before the line is printed)
        out["gathered_rows"] = int(gathered[0].shape[0])   # trajectories of every instance of every rank, on this rank
    if st["qp_iters_last"].size and float(np.mean(st["qp_status_last"] == 0)) == 0.0 and float(np.mean(st["qp_iters_last"])) >= w["P"].qp_iter_max:
        out["note"] = "QPs AT THE ITERATION CAP: this figure times capped interior-point iterations of sub-problems that do not converge (not a solve rate)"
    if warm is not None:
        n_settle, n_timed = warm
        mpc.enable_timing(False)
        for _ in range(n_settle):
            mpc.advance_async()
        mpc.sync()
        mpc.enable_timing(True)
        t0 = time.perf_counter()
        for _ in range(n_timed):
            mpc.advance_async()
        mpc.sync()
        el = time.perf_counter() - t0
        kw, sw = mpc.kernel_times(), mpc.stats()
        roof_w, _ = roofline_objects(w["P"], B, kw, sw, w["P"].sqp_iters, headline=False, key=w.get("key"))
        out["warm"] = {
            "what": f"{n_timed} further SQP iterations after {steps + warmup} cold solves and {n_settle} settling iterations, no reset",
            "value": B * n_timed / el, "unit": "solves/s", "ms_per_step": 1e3 * el / n_timed,
            "qp_converged_fraction": float(np.mean(sw["qp_status_last"] == 0)), "qp_iters_mean": float(np.mean(sw["qp_iters_last"])),
            "constraint_violation_max": float(np.max(sw["constraint_violation"])),
            # the rate of the instances whose QP converged, and how the plans' constraint violation is spread
            "value_converged_subset": float(np.sum(sw["qp_status_last"] == 0)) * n_timed / el,
            "constraint_violation_quantiles": {q: float(np.quantile(sw["constraint_violation"], float(q))) for q in ("0.5", "0.9", "0.99", "1.0")},
            "constraint_violation_of_the_not_converged": (float(np.max(sw["constraint_violation"][sw["qp_status_last"] != 0])) if np.any(sw["qp_status_last"] != 0) else None),
            "constraint_violation_max_of_the_converged": (float(np.max(sw["constraint_violation"][sw["qp_status_last"] == 0])) if np.any(sw["qp_status_last"] == 0) else None),
            "roofline": roof_w,
            "kernel_ms": {"linearize": kw["linearize_ms"], "qp": kw["qp_ms"], "linesearch": kw["linesearch_ms"], "launches": kw["launches"]},
        }
        if "note" in out:
            # the cold launch of this workload times QPs that all end at the iteration cap -- not a solve rate.  The
            # workload's `value` is the rate of the converging regime; the cap-bound launch is kept beside it, labelled
            cold = {k: out[k] for k in ("value", "unit", "ms_per_step", "ms_per_sqp_iter", "steps", "warmup", "qp_converged_fraction", "qp_iters_mean", "roofline", "kernel_ms", "note")}
            out["cold_start_at_the_iteration_cap"] = cold
            wv = out["warm"]
            out.update(value=wv["value"], ms_per_step=wv["ms_per_step"], ms_per_sqp_iter=wv["ms_per_step"], qp_converged_fraction=wv["qp_converged_fraction"],
                       qp_iters_mean=wv["qp_iters_mean"], roofline=wv["roofline"], kernel_ms=wv["kernel_ms"], steps=n_timed,
                       value_converged_subset=wv["value_converged_subset"], constraint_violation_quantiles=wv["constraint_violation_quantiles"])
            out["value_is"] = "the converging regime (`warm`): " + wv["what"]
            del out["note"]
    mpc.close()
    return out


def softened_row_violation(mpc, w):
    """Largest |object-dynamics row| along every instance's plan (the rows configs[3] softens with HPIPM slacks): evaluated by the
    constraint kernel at the plan's own knots (xs_k, us_k), k < N, with each instance's inertial parameters."""
    P = w["P"]
    _, xs, us = mpc.solution()
    B, N = xs.shape[0], P.N
    x = xs[:, :N].reshape(B * N, -1); u = us.reshape(B * N, -1)
    out = mpc.linearize_points(x, u, t=np.tile(P.dt * np.arange(N), B), inst=np.repeat(np.arange(B), N))
    return np.abs(out["g"]).reshape(B, N, -1).max(axis=2)      # [B][N]


def config4_horizon_entry(B_long=256):
    """configs[3] at BASELINE's horizon (N = 20) and at the reference's own horizon for upright_robust
    (upright_robust/config/demos/_base.yaml:62: T = 10 s, N = 100 -- since round 6 on the production kernel too, with the whole-horizon
    arrays in a far array: upr_qp3_cfg::KFAR): how far the plans violate the softened balance rows.  At 2 s for a 2.2 m
    move the rows are traded against the end-effector cost; at 10 s the same move balances."""
    out = {"workload": "configs[3] plans at N = 20 (BASELINE) and N = 100 (the reference's own horizon): violation of the softened object-dynamics rows along the plan",
           "unit": "max |row| (rows normalised as balancing_constraints.cpp:144-151)"}
    for N, B in ((20, 256), (100, B_long)):
        w = config4_workload(B)
        w["P"].N = N
        mpc = make_engine(w)
        mpc.enable_timing(True)
        t0 = time.perf_counter()
        mpc.advance()
        el = time.perf_counter() - t0
        st, kt = mpc.stats(), mpc.kernel_times()
        v = softened_row_violation(mpc, w)
        out[f"N{N}"] = {"batch": B, "qp_kernel": kt["qp_kernel"], "ms_per_solve_of_the_batch": 1e3 * el, "qp_launch_ms": kt["qp_ms"],
                        "qp_converged_fraction": float(np.mean(st["qp_status_last"] == 0)), "qp_iters_mean": float(np.mean(st["qp_iters_last"])),
                        "violation_max": float(v.max()), "violation_mean_over_instances_of_max_over_knots": float(v.max(axis=1).mean()),
                        "violation_max_first_half_of_the_plan": float(v[:, :N // 2].max()), "violation_max_last_quarter": float(v[:, 3 * N // 4:].max())}
        mpc.close()
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_threads():
    """Host threads this process may actually run on: the affinity mask, capped by the cgroup CPU quota (the GPU boxes
    expose all 256 hardware threads of the host but grant a quota of a few CPUs: more threads than that only time-slice)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(w, n_sample):
    """The oracle (oracle/upright_oracle.cpp, -O2 -fopenmp) on the first instances of the same batch, same cold start:
    one thread, then OpenMP over instances on every host thread with at least 8 solves per thread."""
    from oracle.oracle import Oracle
    from upright_amd.sampling import stationary_guess

    P, x0, way = w["P"], w["x0"], w["way"]
    xs0, us0 = stationary_guess(x0, P.N, P.nu)
    O = Oracle(P)
    threads = usable_threads()
    n1 = min(len(x0), 256)   # ~2.7 s on one thread
    tc = time.perf_counter()
    O.solve_batch(0.0, x0[:n1], xs0[:n1], us0[:n1], way_p=way[:n1], nthreads=1)
    dt1 = time.perf_counter() - tc
    n = min(len(x0), max(n_sample, 8 * threads))
    reps = max(1, -(-8 * threads // n))          # fewer instances than 8 per thread: solve the sample several times
    idx = np.tile(np.arange(n), reps)
    O.solve_batch(0.0, x0[idx[:threads]], xs0[idx[:threads]], us0[idx[:threads]], way_p=way[idx[:threads]], nthreads=threads)   # spin the pool up
    tc = time.perf_counter()
    _, _, _, used = O.solve_batch(0.0, x0[idx], xs0[idx], us0[idx], way_p=way[idx], nthreads=threads)
    dtn = time.perf_counter() - tc
    return {
        "value": len(idx) / dtn,
        "unit": "solves/s",
        "cores": used,
        "kind": "port",
        "sample": f"{len(idx)} solves ({len(idx) / used:.1f} per thread) of the first {n} instances of the same batch, same cold start; "
                  f"oracle/upright_oracle.cpp (dense-stage Riccati IPM, -O2 -fopenmp), OpenMP over instances on {used} threads (what the "
                  f"container's CPU quota grants of the {os.cpu_count()} hardware threads) of {cpu_model()}; the reference solver (OCS2 fork + HPIPM) is not available, see BASELINE.md",
        "single_thread_value": n1 / dt1,
        "single_thread_ms_per_solve": 1e3 * dt1 / n1,
        "single_thread_sample": f"first {n1} instances, one thread",
        "cpu_model": cpu_model(),
        "host_threads": os.cpu_count(),
        "usable_threads": threads,
    }


def launch_ranks(args):
    """--gpus N without a launcher: start N ranks (one per GPU) as a child process tree BEFORE this process touches the
    GPU, relay the JSON line, exit with the children's code."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def contract_entry(w, steps, warmup):
    """extra_workloads entry for the SURVEY 8(d) start distribution: converged fraction, status counts, the rate of the
    whole batch and the rate of the converged subset (instances / time of the batch: a batch is as slow as its slowest
    instances, and the infeasible ones run to the iteration cap)."""
    out = time_extra(w, steps, warmup)
    out.pop("gathered", None)
    mpc = make_engine(w)
    mpc.advance()
    st = mpc.stats()
    mpc.close()
    status = st["qp_status_last"].astype(int)
    conv = int(np.sum(status == 0))
    out["qp_status_counts"] = {"converged": conv, "iteration_cap": int(np.sum(status == 1)), "factorisation_broke_down": int(np.sum(status == 2))}
    out["value_converged_subset"] = out["value"] * conv / len(status)
    out["note"] = ("rate of the whole batch, infeasible instances included (they run to the iteration cap and are reported per instance); "
                   "value_converged_subset = converged instances / the same time; the headline's level-tray distribution is upright_amd/sampling.py")
    return out


def _strip(e):
    e.pop("gathered", None); e.pop("u0_gathered", None)
    return e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU")
    ap.add_argument("--cpu-sample", type=int, default=1024, help="instances of the CPU-baseline sample (rank 0, N=1): ~11 core-seconds at the default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the other BASELINE configurations (configs[2], configs[3], configs[4])")
    ap.add_argument("--extra-steps", type=int, default=3)
    ap.add_argument("--closed-loop-ticks", type=int, default=150, help="control periods of the configs[4] closed-loop run")
    ap.add_argument("--only", default=None, choices=["config3", "config4", "config5", "config5s", "contract", "headline_r03", "config4_horizon"],
                    help="run ONE of the other workloads alone and print its entry (the per-workload counter passes of tools/profile_all.sh)")
    ap.add_argument("--dry-run", action="store_true", help="control flow only: gloo on the CPU, stand-in engines with fake solutions; prints a line marked dry_run (never a measurement)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE={world}")

    import torch

    dry = args.dry_run
    device = "cpu" if dry else "cuda"
    if not dry:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus

    import __graft_entry__ as g

    if not dry and not g.LIB.exists():
        g.build()

    engine_devices = []

    def engine_for(w):
        # rank r of a node drives GPU LOCAL_RANK: the engine is created on that device explicitly (upr_set_device), not on
        # whatever torch left current
        if not dry:
            e = make_engine(w, device_index=local_rank)
            assert e.device_index() == local_rank, (e.device_index(), local_rank)
        else:
            sys.path.insert(0, str(ROOT / "tests"))
            from standin import StandInEngine   # (test infrastructure: tests/standin.py)

            Pw = w["P"]
            e = StandInEngine(len(w["x0"]), rank * len(w["x0"]), Pw.N, Pw.nx, Pw.nu, nxf=w["x0"].shape[1], device=local_rank)
        engine_devices.append(e.device_index())
        return e

    if args.only:
        if world != 1 or dry:
            raise SystemExit("bench.py --only runs on one GPU")
        if args.only == "config4_horizon":
            print(json.dumps(config4_horizon_entry()))
            return
        wo = {"config3": lambda: config3_workload(4096), "config4": lambda: config4_workload(1024), "config5": lambda: config5_workload(1024),
              "config5s": lambda: config5_workload(1024, slacks=True), "contract": lambda: contract_workload(1024),
              "headline_r03": lambda: headline_r03_geometry_workload(1024)}[args.only]()
        if args.only.startswith("config5"):
            e = time_closed_loop(wo, args.closed_loop_ticks, device=device, engine=engine_for(wo))
        else:
            e = time_extra(wo, args.extra_steps, 1, warm=(9, 3) if args.only == "config3" else None, engine=engine_for(wo))
        print(json.dumps(_strip(e)))
        return
    w = headline_workload(args.batch, rank, world)
    P, B = w["P"], args.batch
    mpc = engine_for(w)
    elapsed, gathered = rank_main(args, mpc, P, dist=dist, device=device, sync_device=None if dry else torch.cuda.synchronize)
    if gathered is not None:
        assert gathered[0].shape[0] == world * B and bool(torch.isfinite(gathered[0]).all())
    kt = mpc.kernel_times()
    st = mpc.stats()
    if not dry:
        kt = aux_kernel_times(mpc, kt)
    mpc.enable_timing(False)
    mpc.close()

    # the other BASELINE configurations: every rank takes part in the sharded ones (configs[3], configs[4]); rank 0 alone
    # prints.  configs[2] and the SURVEY start distribution are single-GPU workloads (N = 1 only).
    extra = []
    if not args.no_extra:
        if world == 1 and not dry:
            extra.append(time_extra(config3_workload(4096), args.extra_steps, 1, warm=(9, 3)))
        w4 = config4_workload(args.batch if dry else 1024, rank, world)
        extra.append(time_extra(w4, args.extra_steps, 1, dist=dist, device=device, engine=engine_for(w4)))
        w5 = config5_workload(args.batch if dry else 1024, rank, world)
        extra.append(time_closed_loop(w5, args.closed_loop_ticks, dist=dist, device=device, engine=engine_for(w5)))
        if not dry:
            w5s = config5_workload(1024, rank, world, slacks=True)
            extra.append(time_closed_loop(w5s, args.closed_loop_ticks, dist=dist, device=device, engine=engine_for(w5s)))
        if world == 1 and not dry:
            extra.append(contract_entry(contract_workload(1024), args.extra_steps, 1))
            extra.append(time_extra(headline_r03_geometry_workload(1024), 10, 3))
            extra.append(config4_horizon_entry())
        extra = [_strip(e) for e in extra]

    rank_devices = [sorted(set(engine_devices))]
    if world > 1:
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, sorted(set(engine_devices)))
    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        roof, lin = roofline_objects(P, B, kt, st, P.sqp_iters, headline=True, key="headline")
        out = {
            "metric": METRIC,
            "value": B * world * args.steps / elapsed,
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
                "workload": w["name"],
                "batch_per_gpu": B,
                "parallelism": f"instances sharded over {world} rank(s); all-gather of solved trajectories" if world > 1 else "single GPU",
                "qp_converged_fraction": float(np.mean(st["qp_status_last"] == 0)),
                "qp_iters_mean": float(np.mean(st["qp_iters_last"])),
            },
            "roofline": roof,
            "roofline_linearize": lin,
            "kernel_ms": {"linearize": kt["linearize_ms"], "qp": kt["qp_ms"], "linesearch": kt["linesearch_ms"], "launches": kt["launches"],
                          "source": "qp: HIP events on the engine's stream around every fourth QP launch of the timed region (launches[1] of them); linearize, linesearch: "
                                    "10 more solves of the same batch behind the region with events around every kernel (qp there: "
                                    f"{kt.get('aux_qp_ms', 0.0):.4f} ms)"},
            "rank_devices": rank_devices,      # HIP device index of every engine each rank created (rank r: [LOCAL_RANK r])
        }
        if dry:
            out["dry_run"] = True
            out["value"] = 0.0; out["ms_per_step"] = 0.0; out["ms_per_sqp_iter"] = 0.0
            out["data"] = "none: --dry-run exercises the control flow with stand-in engines; no number of this line is a measurement"
        if extra:
            out["extra_workloads"] = extra
        if world == 1 and not args.no_cpu_baseline and not dry:
            out["cpu_baseline"] = cpu_baseline(w, args.cpu_sample)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
