"""CPU tier: `python bench.py --gpus N` must start by itself (no torchrun around it, no WORLD_SIZE in the environment):
it re-executes under torch.distributed.run, one rank per GPU, and rank 0 prints ONE JSON line.  Exercised here without a
GPU through --selftest-launcher (gloo, CPU tensors, a probe matcher whose rows depend on their halo rows -- not a
measurement); the driver's torchrun form of the same command is covered too."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 3, 8])
def test_bench_launches_its_own_ranks(n):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
                        "--selftest-launcher"], capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_json_line(r.stdout)
    assert line["ok"] is True and line["n_gpus"] == n and line["steps"] == 2 and line["warmup"] == 1


def test_bench_under_the_drivers_torchrun_command():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0", "--selftest-launcher"], capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_json_line(r.stdout)
    assert line["ok"] is True and line["n_gpus"] == 2


def test_single_rank_selftest_needs_no_launcher():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--selftest-launcher"], capture_output=True,
                       text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _one_json_line(r.stdout)["n_gpus"] == 1
