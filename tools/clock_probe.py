#!/usr/bin/env python3
"""Does the fill get faster when the GPU has been busy for a while (clock ramp)?  Times single fills
after 0, 20 and 200 back-to-back fills.  usage: clock_probe.py [N] [D]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libstb_amd import capi, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1
T = capi.DeviceTables(N, N, D=D)
a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
T.fill(a); torch.cuda.synchronize()
for busy in (0, 20, 200, 1000):
    time.sleep(0.5)
    for _ in range(busy):
        T.fill(a)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    for i in range(5):
        e[i].record(); T.fill(a)
    e[5].record(); torch.cuda.synchronize()
    print(f"after {busy:4d} fills: " + " ".join(f"{e[i].elapsed_time(e[i+1]):.3f}" for i in range(5)) + " ms", flush=True)
