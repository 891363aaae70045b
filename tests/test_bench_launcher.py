"""bench.py's own launcher: `python bench.py --gpus N` with no launcher in the environment starts a fresh
child (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) BEFORE any GPU call, relays the
child's output and leaves with its exit code.  Checked here without a GPU: the command it builds, that the
child is reached before anything else happens (a stand-in launcher module records what it was given), and the
speed-up fields against a committed one-GPU line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench

    return bench


def test_launcher_command_shape():
    b = _bench()
    cmd = b.launcher_command(4, ["--gpus", "4", "--steps", "7", "--warmup", "2"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]  # the caller's own arguments, unchanged
    assert 1024 < b.free_port() < 65536


def test_self_launch_starts_a_child_and_relays_its_exit_code(tmp_path):
    """a stand-in `torch.distributed.run` on PYTHONPATH records its arguments and environment and exits 7"""
    pkg = tmp_path / "torch" / "distributed"
    pkg.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text("")
    (pkg / "run.py").write_text(
        "import json, os, sys\n"
        f"json.dump({{'argv': sys.argv[1:], 'ipc': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'), 'ws': os.environ.get('WORLD_SIZE')}}, open({str(tmp_path / 'seen.json')!r}, 'w'))\n"
        "print('{\"metric\": \"stand-in\"}')\n"
        "sys.exit(7)\n")
    code = (
        "import sys, os\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        f"os.environ['PYTHONPATH'] = {str(tmp_path)!r}\n"   # only the CHILD sees the stand-in
        "os.environ.pop('WORLD_SIZE', None)\n"
        "sys.argv = ['bench.py', '--gpus', '2', '--steps', '3']\n"
        "bench.main()\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 7, r.stderr[-2000:]
    assert '{"metric": "stand-in"}' in r.stdout      # the child's line reaches our stdout
    seen = json.load(open(tmp_path / "seen.json"))
    a = seen["argv"]
    assert a[a.index("--nproc-per-node") + 1] == "2" and a[-4:] == ["--gpus", "2", "--steps", "3"]
    assert seen["ipc"] == "0" and seen["ws"] is None


def test_under_a_launcher_the_world_size_must_match():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)


def test_speedup_fields_against_the_committed_one_gpu_line():
    b = _bench()
    s = b.speedup_vs_n1(8, 0.8, 0.4, 0.6)
    if s is None:
        pytest.skip("no committed profiles/r0X_bench_n1.json")
    assert s["ranks"] == 8 and s["n1_source"].startswith("profiles/")
    assert s["fill"] == pytest.approx(s["n1_ms"]["fill"] / 0.8)
    assert s["grid_aterms"] == pytest.approx(s["n1_ms"]["grid_aterms"] / 0.4)
