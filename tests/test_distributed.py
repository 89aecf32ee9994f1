"""N > 1 path on CPU: two processes over gloo shard a batch and all-gather 'solved trajectories'
(stand-in arrays: the solve itself needs a GPU), including a ragged split; bench.py's rank functions for the headline
workload, for configs[3] (sharded upright_robust scenarios, all-gather of trajectories) and for configs[4] (sharded goal
sweep in closed loop, all-gather of u_0 per tick) with stand-in engines; and bench.py --gpus 2 --dry-run end to end."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from upright_amd.distributed import all_gather_solutions, shard_range

sys.path.insert(0, str(Path(__file__).resolve().parent))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(total, rank, world)
    n1, nx, N, nu = 21, 27, 20, 21
    # instance b's "solution" is a deterministic function of b so the gather order can be checked
    ids = torch.arange(lo, hi, dtype=torch.float64)
    xs = ids[:, None, None] + torch.zeros(hi - lo, n1, nx, dtype=torch.float64)
    us = -ids[:, None, None] + torch.zeros(hi - lo, N, nu, dtype=torch.float64)
    gx, gu, counts = all_gather_solutions(xs, us)
    ok = gx.shape == (total, n1, nx) and gu.shape == (total, N, nu)
    ok = ok and bool(torch.all(gx[:, 0, 0] == torch.arange(total, dtype=torch.float64)))
    ok = ok and bool(torch.all(gu[:, 3, 5] == -torch.arange(total, dtype=torch.float64)))
    ok = ok and counts == [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    q.put((rank, ok))
    dist.destroy_process_group()


def _run(total):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_all_gather_even_split():
    _run(16)


def test_all_gather_ragged_split():
    _run(7)


from standin import StandInEngine


def _bench_worker(rank, world, port, q):
    import argparse
    import types

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench

    B, n1, nx, N, nu = 5, 21, 27, 20, 21
    P = types.SimpleNamespace(N=N, nx=nx, nu=nu)
    eng = StandInEngine(B, rank * B, N, nx, nu)
    args = argparse.Namespace(gpus=world, steps=3, warmup=1)
    elapsed, (gx, gu) = bench.rank_main(args, eng, P, dist=dist, device="cpu")
    ok = elapsed > 0 and gx.shape == (world * B, n1, nx) and gu.shape == (world * B, N, nu)
    ok = ok and bool(torch.all(gx[:, 0, 0] == torch.arange(world * B, dtype=torch.float64)))
    ok = ok and bool(torch.all(gu[:, 2, 1] == -torch.arange(world * B, dtype=torch.float64)))
    ok = ok and eng.calls == ["reset", "advance"] * 4
    # a launch whose --gpus disagrees with the process group must not run
    try:
        bench.rank_main(argparse.Namespace(gpus=world + 1, steps=1, warmup=0), eng, P, dist=dist, device="cpu")
        ok = False
    except SystemExit:
        pass
    q.put((rank, ok))
    dist.destroy_process_group()


def test_bench_rank_function_two_ranks():
    """bench.py's per-rank loop (warm-up, timed steps, all-gather of solved trajectories through
    upright_amd.distributed.all_gather_solutions, max-over-ranks time) with two gloo processes and stand-in engines."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_bench_self_launch_refuses_mismatched_world(monkeypatch):
    """`--gpus N` under a launcher with another WORLD_SIZE exits non-zero before any GPU call."""
    import subprocess
    import sys
    from pathlib import Path

    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(Path(__file__).resolve().parents[1] / "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr


def _extra_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench

    B = 4
    ok = True
    # configs[3]: rank r owns scenarios [r B, (r + 1) B) of the global sample; per-instance parameters travel with the shard
    w4 = bench.config4_workload(B, rank, world)
    wg = bench.config4_workload(B * world, 0, 1)
    ok = ok and np.array_equal(w4["body_params"], wg["body_params"][rank * B:(rank + 1) * B]) and np.array_equal(w4["x0"], wg["x0"][rank * B:(rank + 1) * B])
    P = w4["P"]
    eng = StandInEngine(B, rank * B, P.N, P.nx, P.nu)
    out = bench.time_extra(w4, 2, 1, dist=dist, device="cpu", engine=eng)
    gx, gu = out["gathered"]
    ok = ok and out["n_gpus"] == world and gx.shape == (world * B, P.N + 1, P.nx) and gu.shape == (world * B, P.N, P.nu)
    ok = ok and bool(torch.all(gx[:, 0, 0] == torch.arange(world * B, dtype=torch.float64)))   # rank order = instance order
    ok = ok and eng.calls == ["reset", "advance"] * 3 and out["exchange"] == "all-gather of solved trajectories"
    # configs[4]: the goal sweep is sharded (rank r owns goals [r B, (r + 1) B) of one circle); per tick only u_0 is gathered
    w5 = bench.config5_workload(B, rank, world)
    w5g = bench.config5_workload(B * world, 0, 1)
    ok = ok and np.allclose(w5["way"], w5g["way"][rank * B:(rank + 1) * B]) and np.allclose(w5["x0"], w5g["x0"][rank * B:(rank + 1) * B])
    P5 = w5["P"]
    eng5 = StandInEngine(B, rank * B, P5.N, P5.nx, P5.nu, nxf=w5["x0"].shape[1])
    ticks = 3
    out5 = bench.time_closed_loop(w5, ticks, dist=dist, device="cpu", engine=eng5)
    u0 = out5["u0_gathered"]
    t_last = 0.01 * ticks                       # (one untimed tick at t = 0, then `ticks` timed ones)
    ok = ok and u0.shape == (world * B, P5.nu) and out5["n_gpus"] == world and out5["exchange"] == "all-gather of u_0 per tick"
    ok = ok and bool(torch.allclose(u0[:, 0], -(torch.arange(world * B, dtype=torch.float64) + t_last)))
    # (two identical runs of the loop: one with per-kernel events for kernel_ms, the timed one without)
    ok = ok and eng5.calls.count("advance") == 2 * (ticks + 1) and eng5.calls.count("obs") == 2 * (ticks + 1) and eng5.calls.count("reset") == 2 and out5["finite"]
    q.put((rank, ok))
    dist.destroy_process_group()


def test_bench_extra_workloads_two_ranks():
    """configs[3] and configs[4] under two gloo ranks: shards are slices of the one global sample, trajectories (u_0 in the
    closed loop) come back in instance order on every rank, every rank solves once per step / tick."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_extra_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_bench_dry_run_two_ranks():
    """`python bench.py --gpus 2 --dry-run`: bench.py launches its two ranks itself, runs the headline loop and both sharded
    extra workloads over gloo with stand-in engines and exits 0 with one line that says it is not a measurement."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(Path(__file__).resolve().parents[1] / "bench.py"), "--gpus", "2", "--dry-run", "--no-cpu-baseline",
                        "--batch", "5", "--steps", "2", "--warmup", "1", "--closed-loop-ticks", "2", "--extra-steps", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["dry_run"] is True and out["n_gpus"] == 2 and out["value"] == 0.0
    # rank 1 ran with LOCAL_RANK = 1 and created every one of its engines for device 1 (the path the 8-GPU run takes)
    assert out["rank_devices"] == [[0], [1]]
    names = [e["workload"][:10] for e in out["extra_workloads"]]
    assert names == ["configs[3]", "configs[4]"] and all(e["n_gpus"] == 2 for e in out["extra_workloads"])


def test_bench_dry_run_eight_ranks():
    """`python bench.py --gpus 8 --dry-run --batch 1024`: the shape of the driver's 8-GPU run (8 x 1024 = BASELINE's 8192 upright_robust
    scenarios, 8192 goals of the thrown-ball sweep) over gloo with stand-in engines.  Every rank creates its engines for device
    LOCAL_RANK = 0 ... 7; the gathered trajectories / first inputs have 8192 rows on rank 0."""
    import json
    import subprocess
    import time

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.time()
    r = subprocess.run([sys.executable, str(Path(__file__).resolve().parents[1] / "bench.py"), "--gpus", "8", "--dry-run", "--no-cpu-baseline",
                        "--batch", "1024", "--steps", "2", "--warmup", "1", "--closed-loop-ticks", "2", "--extra-steps", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["dry_run"] is True and out["n_gpus"] == 8 and out["value"] == 0.0 and out["config"]["batch_per_gpu"] == 1024
    assert out["rank_devices"] == [[i] for i in range(8)]
    e3, e4 = out["extra_workloads"]
    assert e3["n_gpus"] == 8 and e4["n_gpus"] == 8
    assert e3["gathered_rows"] == 8192 and e4["gathered_rows"] == 8192
    assert time.time() - t0 < 120


def _ragged_worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench

    # a global sample that does not divide by the world: shard_range gives the earlier ranks one more; each rank's shard is the slice
    # of the one global sample, and the ragged all-gather returns them in instance order
    lo, hi = shard_range(total, rank, world)
    wg = bench.config4_workload(total, 0, 1)
    x0, bp = wg["x0"][lo:hi], wg["body_params"][lo:hi]
    P = wg["P"]
    eng = StandInEngine(hi - lo, lo, P.N, P.nx, P.nu)
    xs = torch.from_numpy(eng.xs); us = torch.from_numpy(eng.us)
    gx, gu, counts = all_gather_solutions(xs, us)
    ok = counts == [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)] and sum(counts) == total
    ok = ok and gx.shape == (total, P.N + 1, P.nx) and bool(torch.all(gx[:, 0, 0] == torch.arange(total, dtype=torch.float64)))
    ok = ok and len(x0) == hi - lo and len(bp) == hi - lo
    q.put((rank, ok))
    dist.destroy_process_group()


def test_ragged_shards_eight_ranks():
    """8 ranks, 8192 + 5 scenarios: five ranks own 1025, three 1024; the gather trims the padding and keeps instance order."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ragged_worker, args=(r, world, port, 8197, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]
