"""The multi-rank path.

CPU (no GPU here): two gloo ranks shard a discount grid and all-gather one scalar per discount; the
scalars are a stand-in, so this pins libstb_amd/shard.py only (slices, ragged shares, max over ranks).

GPU (-m gpu): the same two ranks share cuda:0 (a one-GPU box; scalars travel over gloo, as in
bench.py's STB_BENCH_SHARE_GPU rehearsal mode).  Each rank fills ITS half of an 8-point discount
grid through the C ABI and evaluates aterms on its half; the gathered vectors are compared with the
CPU oracle.  On a node the same code runs one rank per GPU over RCCL."""
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


def _gpu_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes as C

        import orc
        from libstb_amd import capi

        L = capi.lib()
        assert L.stb_set_device(0) == 0, capi.last_error()
        dev = torch.device("cuda", 0)
        D, N, M = 8, 600, 600
        grid = synth.discount_grid(D)
        sl = shard.my_slice(D, rank, world)
        mine = np.ascontiguousarray(grid[sl])
        T = capi.DeviceTables(N, M, D=len(mine), device=dev)
        T.fill(mine)
        T.status()
        idx = torch.tensor([T.rowoff(N) + M // 2 - 2], device=dev)
        probes = T.tables.index_select(1, idx).reshape(-1)           # log S^N_{M/2} per discount
        full_probe = shard.gather_scalars(probes, D, dist)
        # grid aterms on this rank's discounts
        g = synth.groups(30, 40, N - 1, "wide")
        Mg = max(int(g.t.max()) + 1, 10)
        Ng = max(int(g.n.max()) + 1, Mg)
        h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar),
                                Ng, Mg, len(mine))
        assert h, capi.last_error()
        # the log-posteriors stay on the device: what the gather takes (no copy to the host and back)
        out = np.zeros(len(mine))
        capi.check(L.stb_groups_aterms(h, capi.dp(mine), len(mine), capi.dp(out)))
        d_post = torch.full((len(mine),), float("nan"), dtype=torch.float64, device=dev)
        capi.check(L.stb_groups_aterms_device(h, capi.dp(mine), len(mine), d_post.data_ptr(), capi.stream_ptr()))
        capi.check(L.stb_groups_wait(h))
        torch.cuda.synchronize()
        assert np.array_equal(d_post.cpu().numpy(), out)
        L.stb_groups_free(h)
        full_post = shard.gather_scalars(d_post, D, dist)
        q.put((rank, full_probe.cpu().numpy().tolist(), full_post.cpu().numpy().tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_fill_their_halves_and_gather_on_one_gpu():
    import orc

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    D, N, M = 8, 600, 600
    grid = synth.discount_grid(D)
    want_probe = []
    for a in grid:
        S1, tab = orc.fill_S(float(a), N, M)
        want_probe.append(tab[orc.row_offset(N, M) + M // 2 - 2])
    g = synth.groups(30, 40, N - 1, "wide")
    Mg = max(int(g.t.max()) + 1, 10)
    Ng = max(int(g.n.max()) + 1, Mg)
    L = orc.oracle()
    scratch = np.zeros(synth.cells(Ng, Mg) + Ng)
    want_post = [L.orc_aterms(float(a), g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t),
                              orc.dp(g.bpar), Ng, Mg, orc.dp(scratch)) for a in grid]
    for rank, probe, post in res:
        assert orc.close(probe, want_probe, 1e-10), rank
        assert orc.close(post, want_post, 1e-10), rank
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]      # both ranks hold the same vectors
