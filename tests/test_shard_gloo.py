"""The multi-rank path on CPU: two gloo ranks shard a discount grid, compute a per-discount scalar
locally and all-gather it; both ranks must hold the full vector in grid order.  (On GPUs the same
code runs over RCCL with the scalars produced by the fill / sweep kernels.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from libstb_amd import shard, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, D, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grid = synth.discount_grid(D)
        sl = shard.my_slice(D, rank, world)
        # stand-in for the device result: a deterministic function of the discount
        local = torch.tensor(np.log1p(grid[sl]) * 1e3 + 7.0, dtype=torch.float64)
        full = shard.gather_scalars(local, D, dist)
        t = shard.max_over_ranks(0.5 + rank, torch.device("cpu"), dist)
        q.put((rank, full.numpy().tolist(), t, sl.start, sl.stop))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("D", [64, 7])
def test_two_rank_discount_shard_and_gather(D):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, D, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = (np.log1p(synth.discount_grid(D)) * 1e3 + 7.0).tolist()
    covered = []
    for rank, full, t, lo, hi in res:
        assert full == want
        assert t == 1.5            # max over ranks of 0.5, 1.5
        covered += list(range(lo, hi))
    assert sorted(covered) == list(range(D))


def test_slices_partition_the_grid():
    for D in (1, 7, 8, 64, 65):
        for world in (1, 2, 4, 8):
            idx = []
            for r in range(world):
                s = shard.my_slice(D, r, world)
                idx += list(range(s.start, s.stop))
            assert idx == list(range(D))
    assert shard.counts(64, 8) == [8] * 8
