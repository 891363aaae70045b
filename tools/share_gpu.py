"""Several processes on ONE GPU (what round 5's scratch run found: four ranks x 16 discounts, 725 ms a step): the fused
evaluation of D discounts x 10^6 pairs at N = 10^4 in `world` processes at once, with the one-launch forms kept
(STB_SHARED_GPU=0), with the library's own rule (unset) and with the forms without waits from the start (=1).
usage: python tools/share_gpu.py [world=4] [steps=12] [D=16]      (repo root, GPU box; at most 6 processes may use the card)"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
D = sys.argv[3] if len(sys.argv) > 3 else "16"
for label, env_extra in (("alone, one-launch forms", None), ("STB_SHARED_GPU=0 (one-launch forms kept)", {"STB_SHARED_GPU": "0"}),
                         ("unset (the library's rule)", {}), ("STB_SHARED_GPU=1 (forms without waits)", {"STB_SHARED_GPU": "1"})):
    w = 1 if env_extra is None else world
    env = {k: v for k, v in os.environ.items() if not k.startswith("STB_SHARED")}
    env.update({"STB_SHARED_GPU": "0"} if env_extra is None else env_extra)
    env["SHARED_WORKER_D"] = D
    with tempfile.TemporaryDirectory() as d:
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shared_worker.py"), str(r), str(w), str(steps), os.path.join(d, f"w{r}.json")],
                                  env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True) for r in range(w)]
        errs = [p.communicate(timeout=900)[1] for p in procs]
        print(f"== {label}: {w} process(es) x {D} discounts x 10^6 pairs, N = 10^4, ms per step", flush=True)
        for r, p in enumerate(procs):
            if p.returncode != 0:
                print(f"   rank {r} failed: {errs[r][-300:]}")
                continue
            o = json.load(open(os.path.join(d, f"w{r}.json")))
            print(f"   rank {r}: " + " ".join(f"{x:.1f}" for x in o["ms"]) + f" | slow launches {o['slow_launches']}, forms without waits at the end: {o['shared_mode']}, "
                  f"same bits every step: {o['same_bits_every_step']}, fused give-ups {o['fallbacks']}", flush=True)
            msg = [l for l in errs[r].splitlines() if "libstb_amd:" in l]
            if msg:
                print("      " + msg[0][:300])
