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


def _one_json_line_sized(stdout):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 8192, stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("record", ["r05_d_bench_line.json", "r05_bench_c3_8_ranks_sharing_one_gpu.json"])
def test_compact_line_of_the_largest_committed_records_stays_under_8_kb(record):
    """BENCH_r05.json came back `parsed: null`: the round-5 line was ~20 KB.  bench.py now prints a compact line (the contract
    keys) and writes the full record to bench_details.json; the compaction is a pure function, checked here on the
    largest full records in profiles/ (the one-GPU line of round 5 and an 8-rank line)."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", record)))
    assert len(json.dumps(full)) > 8192
    full["_frame_width"] = 1920
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT == 8192, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "details"):
        assert k in line, k
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "launches", "hbm"):
        assert k in line["roofline"], k
    assert "workload" in line["config"] and "checksum" in line["config"]
    if full["n_gpus"] == 1:
        for k in ("value", "unit", "cores", "kind", "sample", "wall_s", "effective_cores"):
            assert k in line["cpu_baseline"], k
        assert line["bad1_vs_cpu_ref"]["percent"] == full["bad1_vs_cpu_ref"]["percent"]
        assert line["speedup_vs_cpu_baseline"] == full["speedup_vs_cpu_baseline"]
    else:
        assert len(line["rccl"]["ranks"]) == full["n_gpus"] and line["rccl"]["world_size"] == full["n_gpus"]
    assert not any(isinstance(v, str) and len(v) > 260 for v in line.values())
