"""N > 1 path on CPU: two processes over gloo shard a batch and all-gather 'solved trajectories'
(stand-in arrays: the solve itself needs a GPU), including a ragged split."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from upright_amd.distributed import all_gather_solutions, shard_range


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


class _StandInEngine:
    """The methods bench.rank_main calls on the engine, with deterministic 'solutions' (the solve needs a GPU)."""

    def __init__(self, B, lo, n1, nx, N, nu):
        self.B = B
        ids = np.arange(lo, lo + B, dtype=np.float64)
        self.xs = np.ascontiguousarray(ids[:, None, None] + np.zeros((B, n1, nx)))
        self.us = np.ascontiguousarray(-ids[:, None, None] + np.zeros((B, N, nu)))
        self.calls = []

    def reset_async(self): self.calls.append("reset")
    def advance_async(self): self.calls.append("advance")
    def sync(self): pass
    def enable_timing(self, on): pass

    def copy_solution_device(self, xp, up):
        import ctypes

        ctypes.memmove(xp, self.xs.ctypes.data, self.xs.nbytes)
        ctypes.memmove(up, self.us.ctypes.data, self.us.nbytes)


def _bench_worker(rank, world, port, q):
    import argparse
    import types

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench

    B, n1, nx, N, nu = 5, 21, 27, 20, 21
    P = types.SimpleNamespace(N=N, nx=nx, nu=nu)
    eng = _StandInEngine(B, rank * B, n1, nx, N, nu)
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
