"""The path's only exchange -- an all-gather of one log-posterior per discount -- through RCCL itself, on
the one GPU a test box has: torch.distributed with backend "nccl" (= RCCL on ROCm) and world size 1.  No scaling is
measured here (there is nothing to scale); what this proves is that RCCL loads and initialises beside the library, that
the collective takes the device-resident values stb_groups_aterms_device leaves (no copy to the host and back), and that
the stream ordering between the library's stream, torch's current stream and RCCL's holds."""
import os
import socket

import numpy as np
import pytest

import orc
from libstb_amd import capi, shard, synth

pytestmark = pytest.mark.gpu


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_all_gather_of_device_resident_log_posteriors():
    import torch
    import torch.distributed as dist

    L = capi.lib()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        g = synth.groups(100, 100, 1000, "wide")
        M = max(int(g.t.max()) + 1, 10)
        N = max(int(g.n.max()) + 1, M)
        D = 16
        x = np.ascontiguousarray(synth.discount_grid(64)[::4])
        h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), N, M, D)
        assert h, capi.last_error()
        try:
            want = np.zeros(D)
            capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(want)))
            for rep in range(3):
                d_post = torch.full((D,), float("nan"), dtype=torch.float64, device=dev)
                capi.check(L.stb_groups_aterms_device(h, capi.dp(x), D, d_post.data_ptr(), capi.stream_ptr()))
                # queued on torch's stream BEFORE the host waits for the evaluation: only the device-side ordering
                # (the caller's stream waits for the library's event) stands between the collective and stale values
                allpost = shard.gather_scalars(d_post, D, dist, force_collective=True)
                capi.check(L.stb_groups_wait(h))
                torch.cuda.synchronize()
                assert np.array_equal(allpost.cpu().numpy(), want), rep
            # ... and the probe scalars of a batched fill, as bench.py's step gathers them
            T = capi.DeviceTables(2000, 2000, D=4, device=dev)
            a4 = np.ascontiguousarray(x[:4])
            T.fill(a4, capi.FILL_SCALED)
            idx = torch.tensor([T.rowoff(2000) + 1000 - 2], device=dev)
            probes = T.tables.index_select(1, idx).reshape(-1)
            got = shard.gather_scalars(probes, 4, dist, force_collective=True)
            torch.cuda.synchronize()
            assert torch.equal(got, probes) and bool(torch.isfinite(got).all())
        finally:
            L.stb_groups_free(h)
    finally:
        dist.destroy_process_group()
