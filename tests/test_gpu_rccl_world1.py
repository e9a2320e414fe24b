"""GPU tier: the RCCL branch of the strip flow, executed for real.

A 1-GPU box cannot host two RCCL ranks (duplicate devices are refused), so the multi-rank message plan is covered
over gloo (tests/test_strips_gloo.py, tests/test_gpu_strips_multiprocess.py) -- but the code that only the `nccl`
backend takes (device-resident send / receive buffers, `all_gather_into_tensor` on the byte view of the int16 strips,
no host staging: simplestereo_amd/strips.py StripContext.flat_gather / not staged) ran nowhere.  Here one rank
initialises `nccl` (= RCCL on ROCm) at world size 1 and
  * runs StripContext.step for ASW (consistent) and GSW: the result must equal compute() bit for bit;
  * exchanges halo-sized buffers with ITSELF through one batched isend / irecv group (ncclSend / ncclRecv to self):
    the device-to-device point-to-point path of the halo exchange;
  * runs `bench.py --gpus 1` with SSAMD_BENCH_FORCE_DIST=1 on a small configuration and reads its `rccl` object.
Each part runs in its own process under a timeout (an RCCL hang must not take the test session with it)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[1])
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import simplestereo_amd as ss
from simplestereo_amd import strips
from simplestereo_amd.synth import make_pair
res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
for algo, H, W, params in (("asw", 61, 200, dict(winSize=21, maxDisparity=40, consistent=True)),
                           ("asw", 40, 300, dict(winSize=35, maxDisparity=70)),
                           ("gsw", 45, 150, dict(winSize=11, maxDisparity=30))):
    L, R, _ = make_pair(H, W, params["maxDisparity"], 11)
    m = (ss.passive.StereoASW if algo == "asw" else ss.passive.StereoGSW)(**params)
    tL, tR = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
    ctx = strips.StripContext(m, H, W, 0, 1, dev)
    assert ctx.flat_gather and not ctx.staged and ctx.backend == "nccl"
    ctx.enable_timing(True)
    for _ in range(2):
        full = ctx.step(tL, tR, gather=True)
    t = ctx.read_timing()
    want = m.compute(tL, tR)
    res["%%s_%%dx%%d" %% (algo, W, H)] = {"equal": bool(torch.equal(full, want)), "timing": t}
res["p2p_self"] = strips.p2p_self_probe(dev, 10 * 1920 * 3)
dist.barrier(); dist.destroy_process_group()
print("RESULT " + json.dumps(res))
""" % ROOT


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_strip_context_over_nccl_at_world_1():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", WORKER, str(_free_port())], capture_output=True, text=True, timeout=420, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    print(res)
    assert res["backend"] == "nccl" and res["world"] == 1
    cases = [k for k in res if k[:3] in ("asw", "gsw")]
    assert len(cases) == 3 and all(res[k]["equal"] for k in cases), res
    assert all(res[k]["timing"]["steps"] == 2 and res[k]["timing"]["gather_ms"] >= 0 for k in cases)
    assert res["p2p_self"] is True


def _details(env, tmp_path):
    env["SSAMD_BENCH_DETAILS"] = str(tmp_path / "bench_details.json")
    return env


def _read(r, env):
    """(the ONE compact stdout line the driver parses -- asserted < 8 KB --, the full record bench.py wrote beside it)"""
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) < 8192, len(lines[0])       # BENCH_r05.json: a ~20 KB line came back "parsed": null
    line = json.loads(lines[0])
    assert line["details"] == "bench_details.json"
    return line, json.load(open(env["SSAMD_BENCH_DETAILS"]))


def test_bench_forced_distributed_line_at_world_1(tmp_path):
    env = _details(dict(os.environ, SSAMD_BENCH_FORCE_DIST="1", MASTER_PORT=str(_free_port())), tmp_path)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--config",
                        "c2_480p_d64_w35", "--no-others", "--no-cpu-baseline", "--no-e2e"], capture_output=True, text=True, timeout=600,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len([l for l in r.stdout.splitlines() if l.strip()]) == 1, r.stdout
    compact, line = _read(r, env)
    assert compact["rccl"]["backend"] == "nccl" and compact["rccl"]["ranks"][0]["kernel_ms"] > 0
    rc = line["rccl"]
    assert rc["backend"] == "nccl" and rc["world_size"] == 1 and rc["flat_all_gather_into_tensor"] is True
    assert rc["p2p_loopback_probe"] is True
    assert rc["ranks"][0]["strip_rows"] == [0, 480] and rc["ranks"][0]["kernel_ms"] > 0 and rc["ranks"][0]["gather_ms"] >= 0
    # the same frame without the process group: identical checksum
    env.pop("SSAMD_BENCH_FORCE_DIST")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--config",
                         "c2_480p_d64_w35", "--no-others", "--no-cpu-baseline", "--no-e2e"], capture_output=True, text=True, timeout=600,
                        env=env, cwd=ROOT)
    assert r2.returncode == 0, r2.stderr[-3000:]
    plain = json.loads([l for l in r2.stdout.splitlines() if l.strip()][-1])
    assert len(r2.stdout.strip()) < 8192
    assert plain["config"]["checksum"] == line["config"]["checksum"]
    assert "rccl" not in plain


def test_bench_line_of_a_two_rank_run_explains_itself(tmp_path):
    """the N > 1 branches of bench.py -- slowest rank's roofline, per-rank kernel ms, the gathered map against ONE launch over
    the whole frame, the accuracy figure computed THROUGH the strips -- executed with two ranks sharing GPU 0 over gloo
    (SSAMD_BENCH_SHARE_GPU: RCCL refuses two ranks on one device); launched the way the driver launches a scaling run"""
    env = _details(dict(os.environ, SSAMD_BENCH_SHARE_GPU="1", OMP_NUM_THREADS="1"), tmp_path)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SSAMD_BENCH_FORCE_DIST", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--config", "c2_480p_d64_w35"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    compact, line = _read(r, env)
    assert compact["n_gpus"] == 2 and compact["rccl"]["world_size"] == 2 and len(compact["rccl"]["ranks"]) == 2
    assert compact["config"]["checksum_equals_single_gpu"] is True and compact["bad1_vs_cpu_ref"]["percent"] <= 0.5
    # round 6: host time inside StripContext.step per rank (no synchronisation) and the exchange verdict travel in the compact line
    assert 0 < compact["rccl"]["host_step_ms_max"] < 50 and all(r_["host_step_ms"] > 0 for r_ in compact["rccl"]["ranks"])
    assert compact["bad1_vs_cpu_ref"]["differing_pixels"] == 0          # the default path through two strips: the reference's map
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    rc = line["rccl"]
    assert rc["world_size"] == 2 and len(rc["ranks"]) == 2 and "shared_gpu_test_mode" in rc
    assert [r_["strip_rows"] for r_ in rc["ranks"]] == [[0, 240], [240, 480]]
    assert all(r_["kernel_ms"] > 0 and r_["taps"] > 0 and 0 < r_["valu_frac"] < 1 for r_ in rc["ranks"])
    assert rc["kernel_ms_min"] <= rc["kernel_ms_max"]
    assert line["roofline"]["of_rank"] in (0, 1) and line["roofline"]["kernel_ms"] == rc["kernel_ms_max"]
    assert line["config"]["checksum_equals_single_gpu"] is True and line["config"]["checksum_single_gpu"] == line["config"]["checksum"]
    b = line["bad1_vs_cpu_ref"]
    assert "StripContext over 2 ranks" in b["through"] and b["percent"] <= 0.5 and {"W3a", "W3b", "P2a"} <= set(b["cases"])
    # the headline figure is the WHOLE bench frame (tests/golden/full_cases.npz), every pixel counted, through the two strips
    assert b["headline_case"] == "F3p" and b["pixels"] == 1080 * 1920 and b["tie_exclusion"] == "none"
    assert b["cases"]["F4"]["differing_pixels"] == 0
    assert b["cases"]["P2a"]["exact_percent"] >= 99.0


def test_bench_line_carries_the_contract_keys(tmp_path):
    """the line the driver parses: metric / value / unit / steps, `roofline` {bound, achieved, peak, unit, frac, traffic},
    `cpu_baseline` {value, unit, cores, kind, sample}, the accuracy figure on the reference's strips -- on a small
    configuration with a tiny CPU sample so that the test stays short"""
    env = _details(dict(os.environ), tmp_path)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SSAMD_BENCH_FORCE_DIST"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--config", "c2_480p_d64_w35",
                        "--no-others", "--no-e2e", "--cpu-crop-cols", "96"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len([l for l in r.stdout.splitlines() if l.strip()]) == 1, r.stdout
    compact, line = _read(r, env)
    # the compact line alone carries the contract (what the driver sees) ...
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "bad1_vs_cpu_ref"):
        assert k in compact, k
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "launches", "hbm"):
        assert k in compact["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample", "wall_s", "effective_cores"):
        assert k in compact["cpu_baseline"], k
    assert compact["value"] == line["value"] and compact["roofline"]["frac"] == line["roofline"]["frac"]
    assert compact["cpu_baseline"]["value"] == line["cpu_baseline"]["value"] and len(compact["cpu_baseline"]["sample"]) < 200
    assert compact["bad1_vs_cpu_ref"]["percent"] == line["bad1_vs_cpu_ref"]["percent"]
    assert compact["bad1_vs_cpu_ref"]["exact_mode"]["differing_pixels"] == 0
    # ... and the full record beside it keeps everything else
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "bad1_vs_cpu_ref"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1 and line["higher_is_better"] is True
    assert line["value"] > 0 and abs(line["value"] - 480 * 640 * 65 / (line["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * line["value"]
    assert "workload" in line["config"] and line["dtype"] == "f32" and line["data"] == "synthetic"
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert 0 < line["roofline"]["frac"] < 1 and line["roofline"]["kernel_ms"] <= line["ms_per_step"] * 1.02
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert line["cpu_baseline"]["kind"] in ("reference", "port") and line["cpu_baseline"]["value"] > 0
    # the denominator is interpretable: pinned cores, process CPU seconds, effective cores, the per-core rate
    cb = line["cpu_baseline"]
    for k in ("host_threads_visible", "affinity_cpus", "pinned_cpus", "cpu_s", "wall_s", "effective_cores", "taps_per_s_per_effective_core",
              "value_per_effective_core", "all_threads"):
        assert k in cb, k
    assert cb["cores"] == len(cb["pinned_cpus"]) <= 32 and 0 < cb["effective_cores"] <= cb["cores"] * 1.05
    assert line["speedup_per_effective_core"] > line["speedup_vs_cpu_baseline"] > 0
    b = line["bad1_vs_cpu_ref"]
    assert b["percent"] <= 0.5 and b["pixels"] == 1080 * 1920 and b["headline_case"] == "F3p" and b["exact_percent"] >= 99.0
    # ... and the same frame through StereoASW(exact=True): the reference's map itself
    assert b["exact_mode"]["differing_pixels"] == 0 and b["exact_mode"]["pixels"] == 1080 * 1920 and b["exact_mode"]["queue_overflow"] == 0
