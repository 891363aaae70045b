"""Timeline of one halo-block fill (k_fill_hb): the spine's block time per strip, the lag between neighbouring
strips inside a workgroup and across workgroups, and what the tile workers do (wait, load, compute).
usage: python tools/timeline_hb.py N D [out.txt] [M]      (repo root, GPU box; honours STB_HB_* tunables)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from libstb_amd import capi, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1
out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/timeline_hb.txt"
M = int(sys.argv[4]) if len(sys.argv) > 4 else N
raw = out + ".raw"
a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
if os.environ.get("TL_GRID"):
    # the summing form: one fused grid aterms over 10^6 pairs (n < N)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import orc
    L = capi.lib()
    g = synth.groups(1000, 1000, N, "wide")
    M = max(int(g.t.max()) + 1, 10)
    N = max(int(g.n.max()) + 1, M)
    h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), N, M, D)
    assert h, capi.last_error()
    x = np.ascontiguousarray(a)
    o = np.zeros(D)
    for _ in range(3):
        capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(o)))
    os.environ["STB_HB_TIMELINE"] = raw
    capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(o)))
    del os.environ["STB_HB_TIMELINE"]
    L.stb_groups_free(h)
else:
    T = capi.DeviceTables(N, M, D=D)
    for _ in range(3):
        T.fill(a, capi.FILL_HB)
    torch.cuda.synchronize()
    os.environ["STB_HB_TIMELINE"] = raw
    T.fill(a, capi.FILL_HB)
    torch.cuda.synchronize()
    del os.environ["STB_HB_TIMELINE"]
    T.status()

buf = open(raw, "rb").read()
JW, NB, NT, C, P, R, U, Dd = np.frombuffer(buf, dtype=np.int32, count=8)
split, Dd = int(Dd) >> 16, int(Dd) & 0xffff   # (the tiles of a strip's last `split` blocks are listed four times, a quarter of the rows each)
order = np.frombuffer(buf, dtype=np.uint32, count=NT, offset=32)
words = np.frombuffer(buf, dtype=np.uint64, offset=32 + 4 * NT).astype(np.int64)
S = words[: JW * (NB + 2)].reshape(JW, NB + 2)
W = words[JW * (NB + 2):].reshape(NT, 4)
os.remove(raw)
tick = 0.01  # us per wall_clock64 tick (100 MHz)
t0 = S[:, 0][S[:, 0] > 0].min()
UC = U * C
lines = [f"# k_fill_hb timeline, N={N} M={M}, D={D} (table 0 stamped), C={C} P={P}: {JW} strips of {UC} own columns "
         f"({64 - U} halo lanes), {NB} blocks of {R} rows; times in us from the first spine wave's start"]
for j in range(JW):
    s = S[j]
    bl = s[1:NB + 1]
    tt = bl[bl > 0]
    if len(tt) >= 3:
        dt = np.diff(tt) * tick * 1000 / R
        txt = f"row time median {np.median(dt):6.1f} ns (p10 {np.percentile(dt, 10):.1f} p90 {np.percentile(dt, 90):.1f})"
    else:
        txt = ""
    if j < 8 or j % 8 == 0 or j >= JW - 2:
        lines.append(f"strip {j:3d}: start {(s[0] - t0) * tick:8.1f} first block {(tt[0] - t0) * tick if len(tt) else -1:8.1f} end {(s[NB + 1] - t0) * tick:8.1f}  {txt}")
if os.environ.get("TL_BLOCKS"):
    for j in [int(x) for x in os.environ["TL_BLOCKS"].split(",")]:
        bl = S[j, 1:NB + 1]
        tt = bl[bl > 0]
        dt = np.diff(tt) * tick
        lines.append(f"strip {j} block durations (us), first 80: " + " ".join(f"{x:.1f}" for x in dt[:80]))
        lines.append(f"strip {j} block durations: " + ", ".join(f"<{e:.1f}us: {int((dt < e).sum())}" for e in (0.9, 1.2, 1.5, 2, 3, 5, 10, 100)))
if os.environ.get("TL_BINS"):
    # is a slow block slow for everybody at that moment?  mean block duration per 20 us of wall time, a few strips
    js = [int(x) for x in os.environ["TL_BINS"].split(",")]
    tmax = (S[:, NB + 1].max() - t0) * tick
    edges = np.arange(0, tmax + 20, 20.0)
    lines.append("mean block duration (us) per 20 us of wall time; strips " + " ".join(map(str, js)))
    rows_ = []
    for j in js:
        bl = S[j, 1:NB + 1]
        tt = (bl[bl > 0] - t0) * tick
        dt = np.diff(tt)
        idx = np.digitize(tt[:-1], edges)
        rows_.append([dt[idx == k].mean() if (idx == k).any() else float("nan") for k in range(1, len(edges))])
    for k in range(len(edges) - 1):
        lines.append(f"  {edges[k]:6.0f}: " + " ".join(f"{r[k]:5.1f}" for r in rows_))
intra, inter = [], []
for j in range(1, JW):
    both = (S[j, 1:NB + 1] > 0) & (S[j - 1, 1:NB + 1] > 0)
    if both.sum() < 3:
        continue
    lag = (S[j, 1:NB + 1][both] - S[j - 1, 1:NB + 1][both]) * tick
    (inter if j % P == 0 else intra).append(np.median(lag))
if intra:
    lines.append(f"block start of a strip after its left neighbour's, same workgroup (us): median of medians {np.median(intra):.2f}, max {np.max(intra):.2f}")
if inter:
    lines.append(f"... across workgroups (us): median {np.median(inter):.2f}; per hop " + " ".join(f"{x:.1f}" for x in inter))
lines.append(f"spine: first start -> last end {(S[:, NB + 1].max() - t0) * tick:.1f} us; strip 0 alone {(S[0, NB + 1] - S[0, 0]) * tick:.1f} us")
w = W[W[:, 2] > 0]
if len(w):
    ld = (w[:, 1] - w[:, 0]) * tick
    cp = (w[:, 2] - w[:, 1]) * tick
    lines.append(f"workers: {len(w)} tiles of table 0; claimed->inputs loaded median {np.median(ld):.1f} us (p90 {np.percentile(ld, 90):.1f}); "
                 f"compute median {np.median(cp):.1f} us (p10 {np.percentile(cp, 10):.1f} p90 {np.percentile(cp, 90):.1f}); sum of compute {cp.sum() / 1000:.2f} ms")
    lines.append(f"last tile done {(w[:, 2].max() - t0) * tick:.1f} us; last spine end {(S[:, NB + 1].max() - t0) * tick:.1f} us")
    # how long after its record was written was a tile done
    jj = (order & (0x3fff if split else 0xffff)).astype(np.int64)
    bb = (order >> 16).astype(np.int64)
    rec_t = S[jj, 1 + bb]
    okm = (W[:, 2] > 0) & (rec_t > 0)
    late = (W[okm, 2] - rec_t[okm]) * tick
    lines.append(f"tile done after its block's record: median {np.median(late):.1f} us, p90 {np.percentile(late, 90):.1f}, max {late.max():.1f}")
    last = np.argsort(W[:, 2])[-10:][::-1]
    lines.append("the last tiles done (us: claimed, inputs loaded, done; strip, block, quarter; its record written): " + "; ".join(
        f"{(W[i, 0] - t0) * tick:.1f} {(W[i, 1] - t0) * tick:.1f} {(W[i, 2] - t0) * tick:.1f} j{int(jj[i])} b{int(bb[i])} q{(int(order[i]) >> 14) & 3 if split else 0} rec {(rec_t[i] - t0) * tick:.1f}"
        for i in last))
    lines.append(f"distinct (xcc, hw id sans wave) values among workers: {len(set((int(x) >> 4) for x in w[:, 3]))}")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
