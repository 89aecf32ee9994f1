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
