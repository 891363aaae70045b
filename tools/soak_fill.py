#!/usr/bin/env python3
"""Soak test of a one-launch fill form (hb: halo blocks, the default; ck: checkpointed): thousands of fills must give
bit-identical tables, never give up (no fallback to the producer/consumer form), alone and next to a background
load of copies and fp64 matmuls.
usage: python tools/soak_fill.py [seconds] [hb|ck|sf|v|vf]      (repo root, GPU box)
(sf: log S written as float, v / vf: the V table as double / float -- the other output kinds of the halo-block form)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libstb_amd import capi, synth
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
form_name = sys.argv[2] if len(sys.argv) > 2 else "hb"
FORM = {"hb": capi.FILL_HB, "ck": capi.FILL_CK}.get(form_name)


class Kind:
    """a table object of another output kind behind DeviceTables' interface"""
    def __init__(self, N, M, D):
        self.T = capi.DeviceFloatTables(N, M, D=D) if form_name == "sf" else capi.DeviceVTables(N, M, D=D, dtype="f32" if form_name == "vf" else "f64")
        self.tables = self.T.tables
        self.S1 = self.T.S1 if hasattr(self.T, "S1") else torch.zeros(1, device="cuda")

    def fill(self, a, form):
        self.T.fill(a)

    def status(self):
        capi.check(L.stb_fill_status())

L = capi.lib()
cases = [(777, 500, 5), (6000, 900, 2), (10000, 10000, 1), (3000, 3000, 3), (4000, 4000, 8), (10000, 10000, 8), (2000, 2000, 40), (10000, 10000, 16),
         (1000, 1000, 1), (20000, 20000, 1)]
fb0 = L.stb_fill_fallbacks()
total = 0
side = torch.cuda.Stream()
big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
mm = torch.randn(2048, 2048, dtype=torch.float64, device="cuda")
for load in (False, True):
    for N, M, D in cases:
        a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
        if FORM is None and (N < 512 or not L.stb_fill_takes_kind(N, M, D, {'sf': 1, 'v': 2, 'vf': 3}[form_name])):
            continue
        T = capi.DeviceTables(N, M, D=D) if FORM is not None else Kind(N, M, D)
        T.tables.zero_()
        T.fill(a, FORM); torch.cuda.synchronize(); T.status()
        ref = T.tables.clone(); refS1 = T.S1.clone()
        n = 0
        t_case = time.time() + budget / (2 * len(cases))
        while time.time() < t_case:
            for _ in range(10):
                if load:
                    with torch.cuda.stream(side):
                        big[: 128 << 20].copy_(big[128 << 20:])
                        mm2 = mm @ mm
                T.tables.zero_()
                T.fill(a, FORM)
                # (bit patterns: the padding of a V table's rows, right of the diagonal, holds 0/0)
                bits = torch.int32 if T.tables.dtype == torch.float32 else torch.int64
                if not torch.equal(T.tables.view(bits), ref.view(bits)) or not torch.equal(T.S1, refS1):
                    bad = (T.tables.view(bits) != ref.view(bits)).nonzero()
                    print(f"MISMATCH N={N} M={M} D={D} load={load} after {n} fills: {bad.shape[0]} elements differ, first {bad[0].tolist()}", flush=True)
                    sys.exit(1)
                n += 1
            T.status()
        if L.stb_fill_fallbacks() != fb0:
            print(f"FELL BACK N={N} M={M} D={D} load={load}: {capi.last_error()}", flush=True)
            sys.exit(1)
        total += n
        print(f"N={N} M={M} D={D} load={load}: {n} fills identical, none gave up", flush=True)
        del T
print(f"soak ok: {total} {form_name} fills", flush=True)
