#!/usr/bin/env python3
"""bench.py -- throughput of the ASW hot path on MI355X (contract in the task prompt).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Both forms work: started WITHOUT a launcher (no WORLD_SIZE in the environment) and with --gpus N > 1, bench.py
re-executes itself under torch.distributed.run on a free local port -- one rank per GPU over RCCL -- and passes rank 0's
ONE JSON line through.

One "step" = one StereoASW compute of one synthetic rectified pair (BASELINE.json
config 3: 1920x1080, maxDisparity=192, winSize=35, gammaC=5, gammaP=17.5), inputs
already resident in HBM as uint8 BGR.  With N>1 ranks the SAME frame is cut into N
row strips (strong scaling, as north_star asks): every step each rank exchanges
winSize//2 halo rows of both images with its neighbours over RCCL (batched
isend/irecv), runs the kernels on its strip and all-gathers the int16 strips.

Rank 0 prints ONE JSON line.  metric = disparity MPixels/s = H*W*nD / t / 1e6
(nD = maxDisparity - minDisparity + 1), whole job.  Extra objects:
  roofline      the dominant kernel (asw_aggregate_pipe_kernel at the headline configuration) against the bound that BINDS it:
                fp32 VALU issue.  achieved = exact window taps of the launch x 3 lane-ops per
                tap (the irreducible v_mul_f32 w = wL*wR, v_fma_f32 N += w*e, v_fma_f32
                S' += w*(40-e)) / kernel time measured live with HIP events on the launch
                stream; peak = 256 CU x 4 SIMD-32 x 2.4 GHz = 7.86e13 lane-ops/s (157.3 TFLOP/s
                fp32).  "hbm" inside it is the roofline north_star names (34 B/pixel algorithmic
                bytes vs 8 TB/s): three orders of magnitude away from binding.
                `traffic` and `issue` are REPLAYED from committed rocprofv3 PMC passes of the
                same command (their "source" key says which file); everything else is live.
  others        the other BASELINE configurations, timed in the same run after the timed
                region (config 2, config 5 on one GPU, config 3 consistent=True, the class
                default D 0..16, and GSW config 4), each with its own kernel time and VALU fraction.
  cpu_baseline  the reference's own C++ extension (oracle/_ref, kind "reference") or
                the plain-C port (oracle/, kind "port", literal mode) timed on this
                box's host cores on a bounded crop of the same frame with one row per host
                thread (the reference hands out one row per job, _passive.cpp:372-374), so that
                every thread is busy, 512 columns wide so that it stays within ~30 s; plus "hoisted": the plain-C port with the same
                algebraic shortcut as the GPU (right weights evaluated once per pixel, not once
                per candidate), kind "port-hoisted", timed the same way.
  bad1_vs_cpu_ref  the second half of BASELINE's metric: % of pixels whose GPU disparity differs from the CPU
                reference's by more than 1 level.  `percent` is taken on the committed FULL-WIDTH reference strips of
                the headline geometry (tests/golden/wide_cases.npz W3a / W3b: 1920 x 72, D 0..192, win 35, maps
                computed by the unmodified reference, W3b on this very frame); the figure on the narrow cpu_baseline
                crop -- whose candidate sets are truncated by construction -- stays next to it as `crop_percent`.
  e2e_host_arrays  the same operators called the way the reference is called: numpy arrays in, numpy array out
                (H2D + kernels + D2H); never `value`.
  rccl          (N > 1, or SSAMD_BENCH_FORCE_DIST=1 at N = 1) what the distributed step actually ran on: backend,
                observed world size, per-rank device / strip rows / kernel ms / halo-exchange ms / gather ms.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (H, W, maxDisparity, minDisparity, winSize)
    "c3_1080p_d192_w35": (1080, 1920, 192, 0, 35),
    "c1_tsukuba_d16_w15": (288, 384, 16, 0, 15),           # BASELINE config 1: the reference's own example size (synthetic pair here)
    "c2_480p_d64_w35": (480, 640, 64, 0, 35),
    "c5_4k_d256_w35": (2160, 4096, 256, 0, 35),
    "default_1080p_d16_w35": (1080, 1920, 16, 0, 35),      # StereoASW() class defaults (passive.py:59) on a 1080p frame
    "small_1080p_d7_w35": (1080, 1920, 7, 0, 35),          # an even smaller range: the support weights dominate
}
GSW_CONFIGS = {
    # BASELINE config 4: StereoGSW class defaults (winSize 11, gamma 10, fMax 120, iterations 3) at 1080p / D 0..192
    "c4_gsw_1080p_d192_w11": (1080, 1920, 192, 0, 11),
    "default_gsw_1080p_d16_w11": (1080, 1920, 16, 0, 11),      # StereoGSW() class defaults (passive.py:133-134) on a 1080p frame
}
GAMMA_C, GAMMA_P = 5.0, 17.5
GSW_GAMMA, GSW_FMAX, GSW_ITER = 10, 120.0, 3
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_PEAK_LANEOPS = 256 * 4 * 32 * 2.4e9   # 256 CU x 4 SIMD-32 x 2.4 GHz (=157.3 TFLOP/s fp32 FMA)
ALGO_BYTES_PER_PIXEL = 34        # read 2 x 16 B records, write 2 B
GSW_ALGO_BYTES_PER_PIXEL = 10    # read 2 x 4 B packed pixels, write 2 B
# irreducible lane-ops per window tap of the ASW aggregation: v_mul_f32 (w = wL*wR), v_fma_f32 (N += w*e),
# v_fma_f32 (S' += w*(40-e)); unpacking e (v_cvt_f32_ubyteN + v_sub per 8 taps) and addressing come on top
VALU_OPS_PER_TAP = 3
VALU_TAP_INSTRUCTIONS = ["v_mul_f32 w=wL*wR", "v_fma_f32 N+=w*e", "v_fma_f32 S'+=w*(40-e)"]
# GSW: bit-exactness with the reference's fp32 loop dictates an UNFUSED multiply and add per tap, both passes
GSW_OPS_PER_TAP = 2
GSW_TAP_INSTRUCTIONS = ["v_mul_f32 t=w*e", "v_add_f32 cost+=t"]


def count_taps(H, W, win, maxD, minD, row0=0, rows=None):
    """exact number of (pixel, candidate, in-image tap) triples of the left-referenced pass"""
    import numpy as np
    p = win // 2
    rows = H if rows is None else rows
    ys = np.arange(row0, row0 + rows)
    vrows = np.minimum(ys + p, H - 1) - np.maximum(ys - p, 0) + 1           # in-image window rows
    # per (x, d): in-image columns need 0 <= x-d-p+j and x-p+j < W  -> j in [max(0,p-(x-d)), min(win, W+p-x))
    x = np.arange(W)[:, None]
    d = np.arange(minD, maxD + 1)[None, :]
    valid = (x - d) >= 0
    cols = np.minimum(win, W + p - x) - np.maximum(0, p - (x - d))
    cols = np.where(valid, np.maximum(cols, 0), 0)
    return int(vrows.sum()) * int(cols.sum())


def cpu_baseline(cfg, seed, rows_per_thread=1, timeout_s=300, crop_cols=512, repeats=2, pin_cores=32, pinned_rows_per_core=4):
    """Time the reference (and the hoisted plain-C port) on a bounded crop of the same frame, on the host cores.

    The reference hands out ONE image row per job to hardware_concurrency() threads (_passive.cpp:352-355, 372-396).
    Two samples, both cropped to the centre `crop_cols` columns (equal-cost row jobs) and scaled to the full frame by
    the exact tap count:

    * HEADLINE (`value`, `cores`): the process is pinned (os.sched_setaffinity, before the reference is loaded: its
      hardware_concurrency() follows the affinity mask) to `pin_cores` CPUs on distinct physical cores and matches
      `pinned_rows_per_core` row jobs per core, so the one-row-per-thread queue is neither starved nor oversubscribed
      and the denominator has a stated core count.  Timed `repeats` times.
    * `all_threads`: every visible host thread, `rows_per_thread` row jobs each (what rounds 1-4 reported as the
      headline; on these shared 256-thread hosts it measured ~18 core-equivalents).  Timed once.

    Every run records process CPU seconds (RUSAGE_SELF covers the reference's threads), the affinity count and the
    cgroup CPU quota, so `effective_cores` = cpu_s / wall_s says how many cores actually worked."""
    H, W, maxD, minD, win = cfg
    cols = min(W, crop_cols)
    c0 = (W - cols) // 2
    code = r"""
import sys, time, json, os, resource
sys.path.insert(0, %r)
rows, path, mode, c0, cols, pin = int(sys.argv[1]), sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
aff0 = sorted(os.sched_getaffinity(0))
pinned = None
if pin > 0:
    # one CPU per physical core first (SMT siblings share the FMA pipes), then siblings if there are not enough cores
    seen, first, rest = set(), [], []
    for c in aff0:
        try:
            sib = open("/sys/devices/system/cpu/cpu%%d/topology/thread_siblings_list" %% c).read().strip()
        except OSError:
            sib = str(c)
        (rest if sib in seen else first).append(c)
        seen.add(sib)
    pinned = (first + rest)[:pin]
    os.sched_setaffinity(0, pinned)
import numpy as np
from oracle import oracle
from simplestereo_amd.synth import make_pair
H, W, maxD, minD, win, seed = %d, %d, %d, %d, %d, %d
L, R, _ = make_pair(H, W, maxD, seed)
r0 = (H - rows) // 2
a, b = np.ascontiguousarray(L[r0:r0 + rows, c0:c0 + cols]), np.ascontiguousarray(R[r0:r0 + rows, c0:c0 + cols])
ref = oracle.ref_module() if mode == "reference" else None
oracle.asw(a[:2, :64], b[:2, :64], 5, 4, 0, %r, %r)       # load the library outside the timed region
try:
    quota = open("/sys/fs/cgroup/cpu.max").read().strip()
except OSError:
    quota = None
ru0 = resource.getrusage(resource.RUSAGE_SELF)
t = time.time()
if mode == "hoisted":
    d = oracle.asw(a, b, win, maxD, minD, %r, %r, False, hoist=True, nthreads=len(os.sched_getaffinity(0))); kind = "port-hoisted"
elif ref is not None:
    d = ref.computeASW(a, b, win, maxD, minD, %r, %r, False); kind = "reference"
else:
    d = oracle.asw(a, b, win, maxD, minD, %r, %r, False, hoist=False, nthreads=len(os.sched_getaffinity(0))); kind = "port"
dt = time.time() - t
ru1 = resource.getrusage(resource.RUSAGE_SELF)
np.save(path, d)
print(json.dumps({"t": dt, "kind": kind, "r0": int(r0), "cpu_s": (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime),
                  "host_threads_visible": os.cpu_count(), "affinity_cpus": len(aff0), "pinned_cpus": pinned,
                  "threads_used": len(os.sched_getaffinity(0)), "cgroup_cpu_max": quota,
                  "worker_threads": os.cpu_count() if kind == "reference" else len(os.sched_getaffinity(0))}))
""" % (ROOT, H, W, maxD, minD, win, seed, GAMMA_C, GAMMA_P, GAMMA_C, GAMMA_P, GAMMA_C, GAMMA_P, GAMMA_C, GAMMA_P)

    import tempfile
    dump = os.path.join(tempfile.gettempdir(), "ssamd_cpu_ref_%d.npy" % os.getpid())
    dump_a = os.path.join(tempfile.gettempdir(), "ssamd_cpu_all_%d.npy" % os.getpid())
    dump_h = os.path.join(tempfile.gettempdir(), "ssamd_cpu_hoist_%d.npy" % os.getpid())

    def run(rows, mode, path, pin):
        out = subprocess.run([sys.executable, "-c", code, str(rows), path, mode, str(c0), str(cols), str(pin)], capture_output=True,
                             text=True, timeout=timeout_s)      # the reference's queue has an empty()/pop() race: bounded wait
        res = json.loads(out.stdout.strip().splitlines()[-1])
        res["rows"] = rows
        return res

    try:
        navail = len(os.sched_getaffinity(0))
    except AttributeError:
        navail = os.cpu_count() or 1
    visible = os.cpu_count() or 1
    # cgroup CPU quota of this container ("max" or "<quota> <period>" microseconds): more runnable threads than that are
    # throttled, so the pinned sample takes at most that many cores (the GPU boxes of this pool: 16 of 256 host threads)
    quota_cores = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota_cores = float(q) / float(per)
    except (OSError, ValueError):
        pass
    npin = max(1, min(pin_cores, navail, int(quota_cores) if quota_cores and quota_cores >= 1 else pin_cores))
    # The reference spawns std::thread::hardware_concurrency() threads -- on glibc 2.35 that is the number of ONLINE CPUs
    # whatever the affinity mask says (probed: taskset -c 0,1 still reports all) -- and its queue hands a row to every thread
    # that saw the queue non-empty: with fewer rows than threads the threads left over block in SafeQueue::pop for ever
    # (safequeue.hpp:106-114, _passive.cpp:29-32).  So every sample has at least as many rows as there are online CPUs; the
    # pinned sample's threads are then time-sliced on `npin` cores, and effective_cores says what they got.
    rows_pin = min(H, max(pinned_rows_per_core * npin, visible, 8))
    rows_all = min(H, max(rows_per_thread * navail, visible, 8))
    full = count_taps(H, W, win, maxD, minD)
    nD = maxD - minD + 1

    def entry(res):
        rows = res["rows"]
        taps = count_taps(rows, cols, win, maxD, minD)       # the crop is matched as a stand-alone sub-image
        t_full = res["t"] * full / taps                      # per-tap cost is uniform
        used = res["threads_used"]
        eff = res["cpu_s"] / res["t"] if res["t"] > 0 else None
        return {"value": H * W * nD / t_full / 1e6, "unit": "MPixels*disp/s", "cores": used, "kind": res["kind"],
                "host_threads_visible": res["host_threads_visible"], "affinity_cpus": res["affinity_cpus"],
                "pinned_cpus": res["pinned_cpus"], "cgroup_cpu_max": res["cgroup_cpu_max"], "worker_threads": res["worker_threads"],
                "strip_row0": res["r0"], "strip_rows": rows, "strip_col0": c0, "strip_cols": cols,
                "rows_per_thread": rows / float(used), "wall_s": res["t"], "cpu_s": res["cpu_s"],
                "effective_cores": eff, "taps": taps, "taps_per_s": taps / res["t"],
                "taps_per_s_per_effective_core": taps / res["cpu_s"] if res["cpu_s"] > 0 else None,
                "sample": "%dx%d centre crop (%d rows = %.1f row jobs per CPU on %d CPUs%s, %d of %d columns) of the same "
                          "frame, %.3g of the frame's %.4g window taps, %.1f s wall / %.1f CPU-s (%.1f effective cores); scaled to "
                          "the full frame by exact tap count" %
                          (cols, rows, rows, rows / float(used), used,
                           " pinned to %d CPUs on distinct physical cores" % len(res["pinned_cpus"]) if res["pinned_cpus"] else
                           " = every CPU of the affinity mask", cols, W, taps / full, float(full), res["t"], res["cpu_s"], eff or 0.0)}
    try:
        load0 = os.getloadavg()
        runs = [entry(run(rows_pin, "reference", dump, npin)) for _ in range(max(1, repeats))]
        cb = dict(runs[0])
        vals = [r["value"] for r in runs]
        cb["value"] = sum(vals) / len(vals)
        cb["value_min"], cb["value_max"] = min(vals), max(vals)
        cb["runs"] = [{"value": r["value"], "wall_s": r["wall_s"], "cpu_s": r["cpu_s"], "effective_cores": r["effective_cores"]} for r in runs]
        cb["wall_s"] = sum(r["wall_s"] for r in runs)
        cb["cpu_s"] = sum(r["cpu_s"] for r in runs)
        cb["effective_cores"] = cb["cpu_s"] / cb["wall_s"]
        cb["taps_per_s_per_effective_core"] = sum(r["taps"] for r in runs) / cb["cpu_s"]
        # the same rate expressed per core in the metric's unit: what ONE busy core of this host delivers
        cb["value_per_effective_core"] = cb["value"] / cb["effective_cores"]
        cb["cgroup_quota_cores"] = quota_cores
        cb["host_loadavg_before"] = list(load0)
        cb["sample"] += "; timed %d times, value = mean, value_min / value_max = the spread" % len(runs)
        cb["map_file"] = dump
    except Exception as e:      # noqa: BLE001  -- the baseline must never sink the bench line
        return {"value": None, "unit": "MPixels*disp/s", "cores": npin, "host_threads_visible": visible, "kind": "unavailable",
                "sample": repr(e)[:200]}
    try:
        # what rounds 1-4 reported: every visible host thread, one row job each (oversubscribed on a loaded shared host)
        ab = entry(run(rows_all, "reference", dump_a, 0))
        os.remove(dump_a)
        ab["value_per_effective_core"] = ab["value"] / ab["effective_cores"] if ab["effective_cores"] else None
        cb["all_threads"] = ab
    except Exception as e:      # noqa: BLE001
        cb["all_threads"] = {"value": None, "kind": "unavailable", "sample": repr(e)[:200]}
    try:
        # the second comparison BASELINE.md section 3 asks for: a CPU mode with the SAME algebraic shortcut as the GPU
        # kernels (the other image's support weights evaluated once per pixel and window row instead of once per
        # candidate; bit-identical maps, tests/test_oracle_golden.py); same pinning and rows as the headline sample
        hb = entry(run(rows_pin, "hoisted", dump_h, npin))
        import numpy as np
        hb["map_equals_reference_map"] = bool(np.array_equal(np.load(dump), np.load(dump_h)))
        os.remove(dump_h)
        hb["value_per_effective_core"] = hb["value"] / hb["effective_cores"] if hb["effective_cores"] else None
        cb["hoisted"] = hb
    except Exception as e:      # noqa: BLE001
        cb["hoisted"] = {"value": None, "kind": "unavailable", "sample": repr(e)[:200]}
    cb["host_loadavg_after"] = list(os.getloadavg())
    return cb


WEIGHT_OPS = 12     # lane-ops per support weight (3 sub, mul, 2 fma, sqrt, mul, exp2, mul + addressing)


def weight_lane_ops(H, W, win, rows=None):
    """algorithmic lane-ops of the support-weight construction: one weight per (pixel, window cell) and image"""
    return WEIGHT_OPS * 2 * (H if rows is None else rows) * W * win * win


def replayed_issue(config_name, k_ms):
    """VALU instruction count of a committed rocprofv3 PMC pass of this configuration (profiles/valu_<config>.json)
    related to this run's kernel time: issued / useful and the issue rate per SIMD.  None when no pass is committed."""
    vpath = os.path.join(ROOT, "profiles", "valu_%s.json" % config_name)
    if not (os.path.exists(vpath) and k_ms):
        return None
    try:
        vj = json.load(open(vpath))
        n = vj["SQ_INSTS_VALU_per_launch"]
        rate = n / (k_ms * 1e-3) / 1024 / 1e9          # 256 CUs x 4 SIMDs
        out = {"wave_instructions_per_launch": n, "achieved_G_wave_instr_per_s_per_simd": rate,
               "plain_fp32_peak_G_wave_instr_per_s_per_simd": vj["plain_fp32_issue_peak_G_wave_instr_per_s_per_simd"],
               "frac": rate / vj["plain_fp32_issue_peak_G_wave_instr_per_s_per_simd"],
               "source": "profiles/valu_%s.json (builder rocprofv3 --pmc SQ_INSTS_VALU run of this configuration, kernel %s); "
                         "the kernel time it is divided by is this run's" % (config_name, vj.get("kernel", "?"))}
        if vj.get("useful_lane_ops_per_launch"):
            out["issued_over_useful_lane_ops"] = 64.0 * n / vj["useful_lane_ops_per_launch"]
            out["useful_definition"] = vj.get("useful_definition")
        return out
    except Exception:      # noqa: BLE001
        return None


def replayed_counters(config_name, k_ms):
    """HBM traffic and VALU instruction counts of the dominant kernel: NOT measured in this run (rocprofv3 PMC passes
    cannot run inside the timed command) but replayed from the committed passes of the same command; tagged as such."""
    traffic = issue = None
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % config_name)
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = next(v.get("hbm_bytes_per_launch") for k, v in tj.items() if "asw_aggregate" in k)
        except Exception:      # noqa: BLE001
            traffic = None
    vpath = os.path.join(ROOT, "profiles", "valu_%s.json" % config_name)
    if os.path.exists(vpath) and k_ms > 0:
        try:
            vj = json.load(open(vpath))
            rate = vj["SQ_INSTS_VALU_per_launch"] / (k_ms * 1e-3) / 1024 / 1e9          # 256 CUs x 4 SIMDs
            issue = {"wave_instructions_per_launch": vj["SQ_INSTS_VALU_per_launch"],
                     "achieved_G_wave_instr_per_s_per_simd": rate,
                     "plain_fp32_peak_G_wave_instr_per_s_per_simd": vj["plain_fp32_issue_peak_G_wave_instr_per_s_per_simd"],
                     "frac": rate / vj["plain_fp32_issue_peak_G_wave_instr_per_s_per_simd"],
                     "source": "profiles/valu_%s.json (builder rocprofv3 --pmc SQ_INSTS_VALU run of this command, kernel %s); "
                               "the kernel time it is divided by is this run's" % (config_name, vj.get("kernel", "?"))}
        except Exception:      # noqa: BLE001
            issue = None
    return traffic, ("profiles/traffic_%s.json (builder rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this command, "
                     "2 x FETCH_SIZE + WRITE_SIZE per the gfx950 correction)" % config_name) if traffic else None, issue


def time_matcher(matcher, tL, tR, slot, steps=3, warmup=1):
    """(wall ms per step, kernel ms per launch of profile slot `slot`) of matcher.compute on resident tensors.
    Short calls are repeated until the timed region is at least ~30 ms long: three launches of a 0.1 ms kernel
    measure the start of the launch queue, not the kernel (round 2 reported 0.22 ms wall for a 0.14 ms Tsukuba call;
    300 calls in a row take 0.123 ms each, tools/call_overhead.py)."""
    import torch
    from simplestereo_amd import _native
    lib = _native.lib()
    for _ in range(warmup):
        out = matcher.compute(tL, tR)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = matcher.compute(tL, tR)
    torch.cuda.synchronize()
    once = time.perf_counter() - t0
    steps = max(steps, min(300, int(0.03 / max(once, 1e-5))))
    # wall time WITHOUT the library's profiling events (two hipEventRecord per kernel: +10 % on a 0.12 ms Tsukuba call), then
    # the kernel time of the same loop with them
    t0 = time.perf_counter()
    for _ in range(steps):
        out = matcher.compute(tL, tR)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    lib.ssamd_profile_enable(1)
    lib.ssamd_profile_reset()
    for _ in range(steps):
        out = matcher.compute(tL, tR)
    torch.cuda.synchronize()
    ms, launches = _native.profile_read()
    lib.ssamd_profile_enable(0)
    k_ms = ms[slot] / max(1, launches[slot]) * (launches[slot] / float(steps)) if launches[slot] else None
    return wall, k_ms, int(out.to(torch.int64).sum().item())


def others(dev, seed):
    """The BASELINE configurations that are not the bench line, timed in the same run (1 GPU, resident inputs)."""
    import numpy as np
    import torch
    import simplestereo_amd as ss
    from simplestereo_amd import _native
    from simplestereo_amd.synth import make_pair
    res = {}
    cache = {}

    def pair(H, W, maxD):
        if (H, W, maxD) not in cache:
            L, R, _ = make_pair(H, W, maxD, seed)
            cache[(H, W, maxD)] = (torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev))
        return cache[(H, W, maxD)]

    jobs = [("c3_1080p_d192_w35_consistent", "c3_1080p_d192_w35", True), ("c1_tsukuba_d16_w15", "c1_tsukuba_d16_w15", False),
            ("c2_480p_d64_w35", "c2_480p_d64_w35", False),
            ("default_1080p_d16_w35", "default_1080p_d16_w35", False), ("small_1080p_d7_w35", "small_1080p_d7_w35", False),
            ("c5_4k_d256_w35_1gpu", "c5_4k_d256_w35", False)]
    # the same configurations WITHOUT the fp64 tie-break pass (StereoASW(exact=False): the fp32 argmin of rounds 1-5's default path)
    jobs += [("c3_1080p_d192_w35_fp32", "c3_1080p_d192_w35", "fp32"), ("c3_1080p_d192_w35_consistent_fp32", "c3_1080p_d192_w35", "fp32+consistent"),
             ("c2_480p_d64_w35_fp32", "c2_480p_d64_w35", "fp32"), ("default_1080p_d16_w35_fp32", "default_1080p_d16_w35", "fp32"),
             ("c1_tsukuba_d16_w15_fp32", "c1_tsukuba_d16_w15", "fp32"), ("c5_4k_d256_w35_1gpu_fp32", "c5_4k_d256_w35", "fp32")]
    for name, cfgname, consistent in jobs:
        try:
            H, W, maxD, minD, win = CONFIGS[cfgname]
            tL, tR = pair(H, W, maxD)
            exact = not isinstance(consistent, str)       # the default: fp64 tie-break pass on -- the reference's map bit for bit
            consistent = consistent is True or consistent == "fp32+consistent"
            m = ss.passive.StereoASW(winSize=win, maxDisparity=maxD, minDisparity=minD, gammaC=GAMMA_C, gammaP=GAMMA_P,
                                     consistent=consistent, exact=exact)
            wall, k_ms, checksum = time_matcher(m, tL, tR, _native.K_ASW_AGG)
            taps = count_taps(H, W, win, maxD, minD)
            nD = maxD - minD + 1
            wops = weight_lane_ops(H, W, win)
            res[name] = {"matcher": "StereoASW", "H": H, "W": W, "maxDisparity": maxD, "minDisparity": minD, "winSize": win,
                         "consistent": consistent, "ms_per_step": wall, "value": H * W * nD / (wall * 1e-3) / 1e6,
                         "unit": "MPixels*disp/s", "kernel_ms": k_ms, "taps": taps, "checksum": checksum,
                         "valu_frac": VALU_OPS_PER_TAP * taps / (k_ms * 1e-3) / VALU_PEAK_LANEOPS if k_ms else None,
                         # SURVEY 8d also counts the support-weight construction (2 H W win^2 weights x ~12 lane-ops): the larger
                         # share of the work for small disparity ranges, ~1.5 % at D 0..192
                         "weight_lane_ops": wops,
                         "valu_frac_with_weights": (VALU_OPS_PER_TAP * taps + wops) / (k_ms * 1e-3) / VALU_PEAK_LANEOPS if k_ms else None,
                         "issue": replayed_issue(cfgname, k_ms) if not consistent else None,
                         "kernel_form": _native.asw_kernel_form(W, H, win, maxD, minD), "exact": exact}
            if exact:
                res[name].update({"candidates_reevaluated": _native.counter("exact_entries"),
                                  "pixels_flagged": [_native.counter("exact_flagged_left"), _native.counter("exact_flagged_right")],
                                  "queue_overflow": _native.counter("exact_overflow")})
        except Exception as e:      # noqa: BLE001
            res[name] = {"error": repr(e)[:200]}
    for name, (H, W, maxD, minD, win) in GSW_CONFIGS.items():
        try:
            tL, tR = pair(H, W, maxD)
            m = ss.passive.StereoGSW(winSize=win, maxDisparity=maxD, minDisparity=minD, gamma=GSW_GAMMA, fMax=GSW_FMAX,
                                     iterations=GSW_ITER)
            wall, k_ms, checksum = time_matcher(m, tL, tR, _native.K_GSW_AGG)
            taps = 2 * count_taps(H, W, win, maxD, minD)          # left- and right-referenced pass (_passive.cpp:428-548, 551-665)
            nD = maxD - minD + 1
            achieved = GSW_OPS_PER_TAP * taps / (k_ms * 1e-3) if k_ms else None
            res[name] = {"matcher": "StereoGSW (+ left-right check and fill, always on)", "H": H, "W": W, "maxDisparity": maxD,
                         "minDisparity": minD, "winSize": win, "gamma": GSW_GAMMA, "fMax": GSW_FMAX, "iterations": GSW_ITER,
                         "ms_per_step": wall, "value": H * W * nD / (wall * 1e-3) / 1e6, "unit": "MPixels*disp/s",
                         "checksum": checksum, "launch": _native.gsw_geometry(W, H, win, maxD, minD),
                         "roofline": {"bound": "valu", "kernel": "gsw_aggregate_kernel (all launches of a step)", "kernel_ms": k_ms,
                                      "taps": taps, "lane_ops_per_tap": GSW_OPS_PER_TAP, "tap_instructions": GSW_TAP_INSTRUCTIONS,
                                      "achieved": achieved, "peak": VALU_PEAK_LANEOPS, "unit": "lane-ops/s",
                                      "frac": achieved / VALU_PEAK_LANEOPS if achieved else None,
                                      "issue": replayed_issue(name, k_ms),
                                      "hbm": {"algorithmic_bytes_per_step": GSW_ALGO_BYTES_PER_PIXEL * H * W,
                                              "frac": GSW_ALGO_BYTES_PER_PIXEL * H * W / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms else None}}}
        except Exception as e:      # noqa: BLE001
            res[name] = {"error": repr(e)[:200]}
    return res


def bad1_on_reference_strips(dev, rank=0, world=1, consistent=False, config="c3_1080p_d192_w35"):
    """The accuracy half of BASELINE.json's metric.  HEADLINE (`percent`, `exact_percent`, `pixels`): the GPU map of the
    WHOLE bench frame -- make_pair(1080, 1920, 192, seed=1), the frame the timed region runs -- against the map the
    unmodified reference computed for it (tests/golden/full_cases.npz F3p, or F3c with --consistent; generated once in the
    build container by tests/golden/make_golden_full.py), all 2 073 600 pixels, no tie exclusion.  `cases` keeps the
    full-width 72-row strips of rounds 2-4 (wide_cases W3a / W3b), the GSW frame of config 4 (F4, bit-exact bar) and a
    real photograph (photo_cases P2a).  0 s of CPU time.  With world > 1 every ASW case goes through a StripContext over
    all ranks (row strips + RCCL halo exchange + all-gather, the path the timed region ran) -- every rank must call
    this; rank 0 gets the figures."""
    import numpy as np
    import torch
    import simplestereo_amd as ss
    from simplestereo_amd import strips
    from simplestereo_amd.synth import make_pair
    gdir = os.path.join(ROOT, "tests", "golden")
    maps = np.load(os.path.join(gdir, "wide_cases.npz"))
    meta = json.load(open(os.path.join(gdir, "wide_cases.json")))
    pmaps = np.load(os.path.join(gdir, "photo_cases.npz"))
    pmeta = json.load(open(os.path.join(gdir, "photo_cases.json")))
    ppairs = np.load(os.path.join(gdir, "photo_pairs.npz"))
    fmaps, fmeta = None, {}
    if os.path.exists(os.path.join(gdir, "full_cases.npz")):
        fmaps = np.load(os.path.join(gdir, "full_cases.npz"))
        fmeta = json.load(open(os.path.join(gdir, "full_cases.json")))
    head = "F3c" if consistent else "F3p"
    others_f = ("F3c" if head == "F3p" else "F3p", "F4")
    if config == "c5_4k_d256_w35" and not consistent:      # --config c5_4k_d256_w35: the 4096 x 2160 frame against ITS reference map (round 6)
        head, others_f = "F5p", ()
    order = [c for c in (head,) + tuple(others_f) if fmaps is not None and c in fmaps.files] + ["W3a", "W3b", "P2a"]
    cases, frames = {}, {}
    for cid in order:
        if cid.startswith("F"):
            m = fmeta[cid]
            key = tuple(m["frame"])
            if key not in frames:
                frames[key] = make_pair(*key)[:2]
            a, b = frames[key]
            want, what = fmaps[cid], m["recipe"]
        elif cid.startswith("W"):
            m = meta[cid]
            key = tuple(m["frame"])
            if key not in frames:
                frames[key] = make_pair(*key)[:2]
            L, R = frames[key]
            a = np.ascontiguousarray(L[m["row0"]:m["row0"] + m["rows"]])
            b = np.ascontiguousarray(R[m["row0"]:m["row0"] + m["rows"]])
            want, what = maps[cid], m["recipe"]
        else:
            m = pmeta[cid]
            a, b = np.ascontiguousarray(ppairs[m["pair"] + "_L"]), np.ascontiguousarray(ppairs[m["pair"] + "_R"])
            want, what = pmaps[cid], "photograph: tests/golden/photo_pairs.npz %s (reference examples/res/2 lawn pair, rectified, native width)" % m["pair"]
        p = {k: v for k, v in m["params"].items() if k not in ("algo", "frame")}
        gsw = m["params"]["algo"] == "gsw"
        matcher = ss.passive.StereoGSW(**p) if gsw else ss.passive.StereoASW(**p)
        if world > 1 and not gsw:
            rows = a.shape[0]
            q0, q1 = strips.strip_bounds(rows, world, rank)
            ctx = strips.StripContext(matcher, rows, a.shape[1], rank, world, dev)
            d = ctx.step(torch.from_numpy(np.ascontiguousarray(a[q0:q1])).to(dev),
                         torch.from_numpy(np.ascontiguousarray(b[q0:q1])).to(dev), gather=True).cpu().numpy()
        else:
            d = matcher.compute(a, b)
        diff = np.abs(d.astype(np.int32) - want.astype(np.int32))
        cases[cid] = {"percent": 100.0 * float(np.mean(diff > 1)), "exact_percent": 100.0 * float(np.mean(diff == 0)),
                      "pixels": int(diff.size), "bad1_pixels": int(np.count_nonzero(diff > 1)),
                      "differing_pixels": int(np.count_nonzero(diff)), "matcher": "GSW" if gsw else "ASW",
                      "consistent": bool(p.get("consistent", True if gsw else False)), "maxDisparity": p["maxDisparity"],
                      "minDisparity": p["minDisparity"], "recipe": what}
    through = "one launch per case" if world == 1 else "StripContext over %d ranks (row strips, RCCL halo exchange, all_gather)" % world
    exact_mode = fp32_mode = None
    if head in cases and world == 1:
        # the same frame with the fp64 tie-break pass explicitly on (= the default since round 6, DESIGN 4.7: the reference's map
        # itself) and explicitly off (StereoASW(exact=False): the fp32 argmin, the default of rounds 1-5)
        from simplestereo_amd import _native
        for flag in (True, False):
            try:
                m = fmeta[head]
                p = {k: v for k, v in m["params"].items() if k not in ("algo", "frame")}
                a, b = frames[tuple(m["frame"])]
                d = ss.passive.StereoASW(exact=flag, **p).compute(a, b)
                diff = np.abs(d.astype(np.int32) - fmaps[head].astype(np.int32))
                r = {"percent": 100.0 * float(np.mean(diff > 1)), "exact_percent": 100.0 * float(np.mean(diff == 0)),
                     "differing_pixels": int(np.count_nonzero(diff)), "pixels": int(diff.size),
                     "what": "StereoASW(exact=%s) on the whole bench frame against the same reference map (%s)" % (flag, head)}
                if flag:
                    r.update({"candidates_reevaluated": _native.counter("exact_entries"), "queue_overflow": _native.counter("exact_overflow")})
            except Exception as e:      # noqa: BLE001
                r = {"error": repr(e)[:200]}
            if flag:
                exact_mode = r
            else:
                fp32_mode = r
    if head in cases:
        h = cases[head]
        return {"percent": h["percent"], "exact_percent": h["exact_percent"], "pixels": h["pixels"], "bad1_pixels": h["bad1_pixels"],
                "differing_pixels": h["differing_pixels"], "headline_case": head, "tie_exclusion": "none", "exact_mode": exact_mode, "fp32_mode": fp32_mode,
                "cases": cases, "through": through,
                "source": "tests/golden/full_cases.npz %s: the WHOLE frame of this run (config 3: make_pair(1080,1920,192,seed=1), D 0..192; config 5 / F5p: "
                          "make_pair(2160,4096,256,seed=1), D 0..256; win 35, consistent=%s) through the unmodified reference (_passive.cpp via oracle/_ref, "
                          "tests/golden/make_golden_full.py), every pixel counted; cases: the other full-frame maps (F3c/F3p, "
                          "F4 = GSW config 4), the 72-row strips W3a / W3b of earlier rounds and a photograph (P2a)" % (head, consistent)}
    # (no full-frame golden in the tree: the strips of rounds 2-4)
    bad = sum(int(round(c["percent"] * c["pixels"] / 100.0)) for k, c in cases.items() if k.startswith("W"))
    pix = sum(c["pixels"] for k, c in cases.items() if k.startswith("W"))
    exact = sum(c["exact_percent"] * c["pixels"] / 100.0 for k, c in cases.items() if k.startswith("W"))
    return {"percent": 100.0 * bad / pix, "exact_percent": 100.0 * exact / pix, "pixels": pix, "cases": cases, "through": through,
            "source": "tests/golden/wide_cases.npz W3a + W3b: full-width 1920 x 72 strips of the config-3 frames (full_cases.npz absent)"}


def e2e_host_arrays(seed, resident_ms):
    """The operators called like the reference is called -- numpy arrays in, a fresh numpy array out: H2D copies,
    kernels, D2H copy, synchronous (SURVEY 8d "end-to-end per compute()").  Host-array and resident-tensor calls ALTERNATE
    in one loop (same clock and thermal state of the chip for both) and the medians are compared: the overhead of the
    host path is their difference, not the difference to a figure measured minutes earlier."""
    import numpy as np
    import torch
    import simplestereo_amd as ss
    from simplestereo_amd.synth import make_pair
    res = {}
    for name, reps in (("c3_1080p_d192_w35", 7), ("default_1080p_d16_w35", 15), ("c1_tsukuba_d16_w15", 40)):
        H, W, maxD, minD, win = CONFIGS[name]
        L, R, _ = make_pair(H, W, maxD, seed)
        tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
        m = ss.passive.StereoASW(winSize=win, maxDisparity=maxD, minDisparity=minD, gammaC=GAMMA_C, gammaP=GAMMA_P)
        for _ in range(2):
            out = m.compute(L, R)
            m.compute(tL, tR)
        torch.cuda.synchronize()
        th, tr = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = m.compute(L, R)
            th.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            m.compute(tL, tR)
            torch.cuda.synchronize()
            tr.append(time.perf_counter() - t0)
        t, t_res = float(np.median(th)), float(np.median(tr))
        nD = maxD - minD + 1
        res[name] = {"ms_per_step": t * 1e3, "value": H * W * nD / t / 1e6, "unit": "MPixels*disp/s",
                     "bytes_h2d": 2 * H * W * 3, "bytes_d2h": H * W * 2, "checksum": int(out.astype(np.int64).sum()),
                     "resident_same_loop_ms": t_res * 1e3, "overhead_vs_resident_ms": (t - t_res) * 1e3,
                     "resident_ms_per_step_timed_region": resident_ms.get(name)}
    return res


class _ProbeMatcher:
    """Launcher self-test only (--selftest-launcher, CPU + gloo, tests/test_bench_launcher_cpu.py): NOT a stereo matcher
    and never part of a measurement -- a row-window checksum whose value at row y depends on the input rows y-pad ..
    y+pad, so that strips + halo exchange + gather reproduce the whole-frame result only if the plumbing is right."""
    winSize = 7

    def _compute_device(self, t1, t2, out_row0=0, out_rows=None):
        import torch
        H = int(t1.shape[0])
        rows = H - out_row0 if out_rows is None else int(out_rows)
        a = t1[:, :, 0].to(torch.int64) + 2 * t2[:, :, 1].to(torch.int64)
        c = torch.cat([torch.zeros((1, a.shape[1]), dtype=torch.int64), a.cumsum(0)], 0)
        p = self.winSize // 2
        y = torch.arange(out_row0, out_row0 + rows)
        lo, hi = (y - p).clamp(min=0), (y + p + 1).clamp(max=H)
        return ((c[hi] - c[lo]) % 32749).to(torch.int16)


def selftest_launcher(args):
    """ranks on CPU over gloo: StripContext.step == the whole frame; rank 0 prints ONE JSON line"""
    import numpy as np
    import torch
    import torch.distributed as dist
    from simplestereo_amd import strips
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W = 37, 50
    rng = np.random.default_rng(5)
    L = torch.from_numpy(rng.integers(0, 256, (H, W, 3), dtype=np.uint8))
    R = torch.from_numpy(rng.integers(0, 256, (H, W, 3), dtype=np.uint8))
    m = _ProbeMatcher()
    r0, r1 = strips.strip_bounds(H, world, rank)
    ctx = strips.StripContext(m, H, W, rank, world, "cpu")
    for _ in range(args.warmup + args.steps):
        out = ctx.step(L[r0:r1].contiguous(), R[r0:r1].contiguous(), gather=True)
    ok = bool(torch.equal(out, m._compute_device(L, R)))
    oks = [None] * world
    dist.all_gather_object(oks, ok)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        return json.dumps({"metric": "launcher selftest (CPU, gloo, probe matcher -- not a measurement)", "ok": all(oks),
                           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "backend": "gloo"})
    return None


def relaunch_under_torchrun(n):
    """bench.py --gpus N started without a launcher: run N ranks of this very command under torch.distributed.run
    (one process per GPU, RCCL) on a free local port; rank 0's JSON line is the only thing on stdout."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what the host driver supports (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


LINE_LIMIT = 8192       # bytes: the driver's parser gave up on the ~20 KB line of round 5 (BENCH_r05.json "parsed": null)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(full):
    """The ONE stdout line: the contract keys only (the prompt's bench contract + roofline + cpu_baseline + the accuracy
    half of the metric), every prose field cut to a short sentence.  Everything else lives in bench_details.json."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                        "vs_baseline", "dtype", "data"))
    cfg = full.get("config", {})
    line["config"] = _pick(cfg, ("workload", "parallelism", "checksum", "checksum_equals_single_gpu"))
    rf = full.get("roofline", {})
    line["roofline"] = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "kernel_ms", "launches", "traffic",
                                  "taps_per_launch", "lane_ops_per_tap", "of_rank"))
    line["roofline"]["hbm"] = _pick(rf.get("hbm", {}), ("achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch"))
    if isinstance(rf.get("issue"), dict):
        line["roofline"]["issued_over_useful_lane_ops"] = rf["issue"].get("issued_over_useful_lane_ops")
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, ("value", "unit", "cores", "kind", "wall_s", "effective_cores", "cgroup_cpu_max", "value_min", "value_max"))
        if cb.get("value"):
            c["sample"] = "%dx%d centre crop of the same frame (%d rows, %d of %d columns), timed %d x, scaled to the frame by exact tap count" % (
                cb.get("strip_cols", 0), cb.get("strip_rows", 0), cb.get("strip_rows", 0), cb.get("strip_cols", 0),
                full.get("_frame_width", 0), len(cb.get("runs", [])) or 1)
        else:
            c["sample"] = str(cb.get("sample"))[:200]
        line["cpu_baseline"] = c
    b1 = full.get("bad1_vs_cpu_ref")
    if isinstance(b1, dict):
        b = _pick(b1, ("percent", "exact_percent", "pixels", "bad1_pixels", "differing_pixels", "headline_case", "crop_percent"))
        if isinstance(b1.get("exact_mode"), dict):
            b["exact_mode"] = _pick(b1["exact_mode"], ("percent", "differing_pixels", "candidates_reevaluated", "queue_overflow", "error"))
        if isinstance(b1.get("fp32_mode"), dict):
            b["fp32_mode"] = _pick(b1["fp32_mode"], ("percent", "differing_pixels", "error"))
        if isinstance(b1.get("crop_fp32_mode"), dict):
            b["crop_fp32_mode"] = _pick(b1["crop_fp32_mode"], ("percent", "exact_percent"))
        if b1.get("percent") is None and "source" in b1:
            b["source"] = str(b1["source"])[:200]
        line["bad1_vs_cpu_ref"] = b
    for k in ("speedup_vs_cpu_baseline", "speedup_vs_cpu_baseline_cores", "speedup_per_effective_core", "default_mode", "fp32_ms_per_step",
              "exact_overhead_percent", "exact_pass_ms"):
        if k in full:
            line[k] = full[k]
    rc = full.get("rccl")
    if isinstance(rc, dict):
        r = _pick(rc, ("backend", "world_size", "nccl_version", "halo_rows_per_side", "halo_message_bytes", "kernel_ms_min",
                       "kernel_ms_max", "host_step_ms_max", "exchange_hidden", "shared_gpu_test_mode"))
        ranks = [x for x in rc.get("ranks", []) if x]
        r["ranks"] = [_pick(x, ("rank", "device", "strip_rows", "kernel_ms", "halo_exchange_exposed_ms", "gather_ms", "host_step_ms"))
                      for x in ranks[:8]]
        line["rccl"] = r
    line["details"] = "bench_details.json"
    return line


def emit(full):
    """Write the complete record to bench_details.json (repo root and gpurun_out/) and to stderr; return the compact line."""
    blob = json.dumps({k: v for k, v in full.items() if not k.startswith("_")}, indent=1, sort_keys=True)
    paths = [os.path.join(ROOT, "bench_details.json"), os.path.join(ROOT, "gpurun_out", "bench_details.json")]
    if os.environ.get("SSAMD_BENCH_DETAILS"):           # tests point this at a scratch file
        paths = [os.environ["SSAMD_BENCH_DETAILS"]]
    for path in paths:
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                f.write(blob + "\n")
        except OSError:
            pass
    sys.stderr.write("bench_details: " + json.dumps(full) + "\n")
    sys.stderr.flush()
    line = compact_line(full)
    result = json.dumps(line)
    if len(result) >= LINE_LIMIT:        # never hand the driver a line it cannot parse: drop the optional blocks, keep the contract
        for k in ("rccl", "bad1_vs_cpu_ref"):
            line.pop(k, None)
            result = json.dumps(line)
            if len(result) < LINE_LIMIT:
                break
    assert len(result) < LINE_LIMIT, len(result)
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c3_1080p_d192_w35", choices=sorted(CONFIGS))
    ap.add_argument("--consistent", action="store_true")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--fp32", action="store_true",
                    help="time StereoASW(exact=False): the fp32 argmin without the fp64 tie-break pass (the default path of rounds 1-5); the line says so")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip the other BASELINE configurations (extra JSON key `others`)")
    ap.add_argument("--cpu-rows-per-thread", type=int, default=1,
                    help="cpu_baseline sample height in rows per host thread (the reference schedules one row per job)")
    ap.add_argument("--cpu-crop-cols", type=int, default=512,
                    help="cpu_baseline sample width (centre columns); 0 = full width (minutes of CPU time at 1080p)")
    ap.add_argument("--cpu-repeats", type=int, default=2, help="how many times the (pinned) cpu_baseline sample of the reference is timed")
    ap.add_argument("--cpu-pin-cores", type=int, default=32,
                    help="cpu_baseline headline: pin the reference to this many CPUs on distinct physical cores (min with the affinity mask)")
    ap.add_argument("--cpu-pinned-rows-per-core", type=int, default=4, help="row jobs per pinned core in the headline cpu_baseline sample")
    ap.add_argument("--with-alternate", action="store_true",
                    help="also time the opt-in alternate-rows mode after the timed region (extra JSON key)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-array (PCIe-inclusive) timings")
    ap.add_argument("--no-bad1", action="store_true",
                    help="skip the accuracy figure on the reference strips (profiling runs: keeps the kernel statistics to the timed launches)")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU + gloo self-test of the rank launcher and the strip plumbing with a probe matcher (no GPU, "
                         "not a measurement): tests/test_bench_launcher_cpu.py")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(relaunch_under_torchrun(args.gpus))

    # stdout must carry exactly ONE JSON line.  RCCL (NCCL_DEBUG=VERSION on the GPU boxes) and other
    # native libraries write banners / warnings to C stdout, flushed at exit: keep the real stdout in a
    # private descriptor and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    if args.selftest_launcher:
        os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("RANK", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
        result = selftest_launcher(args)
        sys.stdout.flush()
        if result is not None:
            os.write(real_stdout, (result + "\n").encode())
        return

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world                  # the launcher's world size is authoritative
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # TEST HOOK (tests/test_gpu_rccl_world1.py, never a measurement): SSAMD_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and
    # moves the messages over gloo through host staging -- RCCL refuses two ranks on one device -- so that the N > 1
    # branches of this file (per-rank roofline, map equality with one GPU, accuracy through the strips) can be executed
    # on a box with ONE GPU.  The line says so (`rccl.backend` = "gloo", `shared_gpu_test_mode`).
    share_gpu = os.environ.get("SSAMD_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("SSAMD_BENCH_FORCE_DIST") == "1"   # FORCE: exercise the RCCL path at N=1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL's point-to-point kernels of the halo exchange on high-priority streams: they have to get onto a GPU that the interior
        # rows' aggregation kernel keeps full (one 12-wave workgroup per CU) -- at normal priority they may only be dispatched when
        # that kernel has no workgroup left to place, i.e. at its end (measured with the loopback harness, tools/strip_host_cost.py)
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import simplestereo_amd as ss
    from simplestereo_amd import _native, strips
    from simplestereo_amd.synth import make_pair

    cfg = CONFIGS[args.config]
    H, W, maxD, minD, win = cfg
    nD = maxD - minD + 1
    L, R, _ = make_pair(H, W, maxD, args.seed)          # same bytes on every rank
    r0, r1 = strips.strip_bounds(H, world, rank)
    ownL = torch.from_numpy(np.ascontiguousarray(L[r0:r1])).to(dev)
    ownR = torch.from_numpy(np.ascontiguousarray(R[r0:r1])).to(dev)
    matcher = ss.passive.StereoASW(winSize=win, maxDisparity=maxD, minDisparity=minD, gammaC=GAMMA_C, gammaP=GAMMA_P,
                                   consistent=args.consistent, exact=False if args.fp32 else "auto")

    strip_ctx = strips.StripContext(matcher, H, W, rank, world, dev) if use_dist else None
    p2p_loopback = None
    if use_dist and world == 1:
        # one GPU cannot host a second RCCL rank: the device-to-device point-to-point path of the halo exchange is
        # exercised with this rank as its own peer (ncclSend / ncclRecv to self in one group), outside the timed region
        try:
            p2p_loopback = strips.p2p_self_probe(dev, win // 2 * W * 3)
        except Exception as e:      # noqa: BLE001
            p2p_loopback = "failed: " + repr(e)[:160]

    def step():
        if not use_dist:
            return matcher.compute(ownL, ownR)
        return strip_ctx.step(ownL, ownR, gather=True)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    lib = _native.lib()
    for _ in range(args.warmup):
        out = step()
    fence()
    lib.ssamd_profile_enable(1)
    lib.ssamd_profile_reset()
    if strip_ctx is not None:
        strip_ctx.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    ms, launches = _native.profile_read()
    lib.ssamd_profile_enable(0)
    rccl = None
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if share_gpu else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # what every rank actually ran on, collected on rank 0
        phases = strip_ctx.read_timing() or {}
        strip_ctx.enable_timing(False)
        my_k = ms[_native.K_ASW_AGG] / max(1, args.steps)          # per STEP: the overlapped strip step runs two aggregation launches (interior rows, border bands)
        my_taps = count_taps(H, W, win, maxD, minD, r0, r1 - r0) if r1 > r0 else 0
        mine = {"rank": rank, "device": int(torch.cuda.current_device()), "device_name": torch.cuda.get_device_name(dev),
                "strip_rows": [r0, r1], "halo_rows": [strip_ctx.h0, strip_ctx.h1],
                "kernel_ms": my_k, "taps": my_taps,
                "valu_frac": VALU_OPS_PER_TAP * my_taps / (my_k * 1e-3) / VALU_PEAK_LANEOPS if my_k > 0 else None,
                "halo_exchange_ms": phases.get("exchange_ms"), "kernels_phase_ms": phases.get("kernels_ms"),
                "overlapped": phases.get("overlapped"), "interior_rows": strip_ctx.interior, "border_rows": strip_ctx.top + strip_ctx.bot,
                "interior_ms": phases.get("interior_ms"), "border_ms": phases.get("border_ms"),
                "halo_exchange_exposed_ms": phases.get("exchange_exposed_ms"),
                "aggregation_launches_per_step": launches[_native.K_ASW_AGG] / float(max(1, args.steps)),
                "gather_ms": phases.get("gather_ms"), "messages_sent": len(strip_ctx.sends), "messages_received": len(strip_ctx.recvs),
                # CPU time inside StripContext.step per step (no synchronisation): must stay well below the strip's kernel time
                "host_step_ms": phases.get("host_step_ms")}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        rccl = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "flat_all_gather_into_tensor": bool(strip_ctx.flat_gather),
                "halo_rows_per_side": strip_ctx.pad, "halo_message_bytes": strip_ctx.pad * W * 3,
                "gather_bytes_per_rank": strip_ctx.rows_max * W * 2, "p2p_loopback_probe": p2p_loopback, "ranks": per_rank}
        if share_gpu:
            rccl["shared_gpu_test_mode"] = "every rank on GPU 0, messages over gloo through host staging: NOT a scaling measurement"
        try:
            rccl["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:      # noqa: BLE001
            pass
        ks = [r["kernel_ms"] for r in per_rank if r and r.get("kernel_ms")]
        if ks:
            rccl["kernel_ms_min"], rccl["kernel_ms_max"] = min(ks), max(ks)
        hs = [r["host_step_ms"] for r in per_rank if r and r.get("host_step_ms") is not None]
        if hs:
            rccl["host_step_ms_max"] = max(hs)
        ex = [r.get("halo_exchange_exposed_ms") for r in per_rank if r and r.get("overlapped")]
        if ex and all(v is not None for v in ex):
            # the halo exchange ran under the interior rows on every rank (what is left exposed is below 50 us)
            rccl["exchange_hidden"] = bool(max(ex) < 0.05)
            rccl["exchange_exposed_ms_max"] = max(ex)
    checksum = int(out.to(torch.int64).sum().item())
    bad1_dist = None
    if world > 1 and not args.no_bad1:
        # the accuracy half of the metric THROUGH the distributed path (every rank takes part; rank 0 keeps the figures)
        try:
            bad1_dist = bad1_on_reference_strips(dev, rank, world, consistent=args.consistent, config=args.config)
        except Exception as e:      # noqa: BLE001
            bad1_dist = {"percent": None, "source": repr(e)[:200]}

    if rank == 0:
        per_step = dt / args.steps
        k_ms = ms[_native.K_ASW_AGG] / max(1, args.steps if use_dist else launches[_native.K_ASW_AGG])       # this rank's strip, per step
        rows_here = r1 - r0
        algo_bytes = ALGO_BYTES_PER_PIXEL * rows_here * W
        taps_here = count_taps(H, W, win, maxD, minD, r0, rows_here)
        achieved_gbs = algo_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None
        achieved_ops = VALU_OPS_PER_TAP * taps_here / (k_ms * 1e-3) if k_ms > 0 else None
        geom = _native.asw_geometry(W, rows_here, win, maxD, minD)
        geom.update(_native.asw_kernel_form(W, rows_here, win, maxD, minD))
        traffic, traffic_source, issue = replayed_counters(args.config, k_ms) if world == 1 else (None, None, None)
        line = {
            "_frame_width": W,
            "metric": "disparity MPixels/s (H*W*nDisp per second)",
            "value": H * W * nD / per_step / 1e6,
            "unit": "MPixels*disp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": per_step * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s: ASW %dx%d maxDisparity=%d minDisparity=%d winSize=%d gammaC=%g gammaP=%g consistent=%s, "
                                   "seeded synthetic rectified pair resident in HBM" %
                                   (args.config, W, H, maxD, minD, win, GAMMA_C, GAMMA_P, bool(args.consistent)),
                       "parallelism": "1 GPU, whole frame" if world == 1 else "row strips x%d, RCCL halo exchange + all_gather" % world,
                       "launch": geom, "checksum": checksum},
            # the bound that binds: fp32 VALU issue (no MFMA on this path, as north_star prescribes; HBM is ~3 orders
            # of magnitude away, kept as roofline.hbm because north_star names it)
            "roofline": {"bound": "valu",
                         "kernel": ("asw_aggregate_wave_kernel" if geom.get("wave_kernel") else
                                    "asw_aggregate_pipe_kernel" if geom.get("phase_shifted") else "asw_aggregate_kernel"),
                         "achieved": achieved_ops, "peak": VALU_PEAK_LANEOPS,
                         "unit": "lane-ops/s", "frac": (achieved_ops / VALU_PEAK_LANEOPS) if achieved_ops else None,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel_ms": k_ms, "launches": launches[_native.K_ASW_AGG],
                         "taps_per_launch": taps_here, "lane_ops_per_tap": VALU_OPS_PER_TAP,
                         "tap_instructions": VALU_TAP_INSTRUCTIONS,
                         "peak_definition": "256 CU x 4 SIMD-32 x 2.4 GHz (MI355X_MICROARCH.md: v_fma_f32 wave64 = 2 cycles) = 157.3 TFLOP/s fp32",
                         "issue": issue,
                         "hbm": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": (achieved_gbs / HBM_PEAK_GBS) if achieved_gbs else None,
                                 "algorithmic_bytes_per_launch": algo_bytes,
                                 "note": "the roofline north_star names; not binding for this kernel"}},
            "kernels_ms_per_step": {_native.lib().ssamd_kernel_name(i).decode(): ms[i] / args.steps for i in range(_native.K_COUNT) if launches[i]},
        }
        if rccl is not None:
            line["rccl"] = rccl
        if world > 1:
            # the slowest rank's strip against the roofline (a step ends when the slowest strip does), and the strips'
            # map against ONE launch over the whole frame on rank 0's GPU (outside the timed region)
            slow = max((r for r in rccl["ranks"] if r and r.get("kernel_ms")), key=lambda r: r["kernel_ms"], default=None)
            if slow is not None:
                line["roofline"].update({"frac": slow["valu_frac"], "kernel_ms": slow["kernel_ms"], "taps_per_launch": slow["taps"],
                                         "achieved": VALU_OPS_PER_TAP * slow["taps"] / (slow["kernel_ms"] * 1e-3),
                                         "of_rank": slow["rank"], "note": "the slowest rank's strip"})
            try:
                single = matcher.compute(torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev))
                line["config"]["checksum_single_gpu"] = int(single.to(torch.int64).sum().item())
                line["config"]["checksum_equals_single_gpu"] = bool(torch.equal(single, out.to(single.device)))
            except Exception as e:      # noqa: BLE001
                line["config"]["checksum_equals_single_gpu"] = repr(e)[:160]
            if bad1_dist is not None:
                line["bad1_vs_cpu_ref"] = bad1_dist
        line["default_mode"] = ("exact: StereoASW(exact=\"auto\") -- near-ties of every winner selected in the aggregation kernel and re-decided in "
                                "fp64 in the reference's arithmetic (DESIGN 4.7); the timed region runs this") if not args.fp32 else \
                               "fp32 argmin only (--fp32: StereoASW(exact=False)); NOT the default path"
        if launches[_native.K_ASW_EXACT]:
            line["exact_pass_ms"] = ms[_native.K_ASW_EXACT] / max(1, args.steps)
        if world == 1 and not use_dist and not args.fp32:
            # the same frame WITHOUT the tie-break pass (the default of rounds 1-5), alternating with the default so that both see the
            # same clocks: what the reference's own map costs
            try:
                m32 = ss.passive.StereoASW(winSize=win, maxDisparity=maxD, minDisparity=minD, gammaC=GAMMA_C, gammaP=GAMMA_P,
                                           consistent=args.consistent, exact=False)
                m32.compute(ownL, ownR)
                torch.cuda.synchronize()
                t32, t64 = [], []
                for _ in range(max(3, min(10, args.steps))):
                    ta = time.perf_counter(); m32.compute(ownL, ownR); torch.cuda.synchronize(); t32.append(time.perf_counter() - ta)
                    ta = time.perf_counter(); matcher.compute(ownL, ownR); torch.cuda.synchronize(); t64.append(time.perf_counter() - ta)
                line["fp32_ms_per_step"] = float(np.median(t32)) * 1e3
                line["exact_ms_per_step_same_loop"] = float(np.median(t64)) * 1e3
                line["exact_overhead_percent"] = 100.0 * (float(np.median(t64)) / float(np.median(t32)) - 1.0)
            except Exception as e:      # noqa: BLE001
                line["fp32_ms_per_step"] = None
                line["exact_overhead_error"] = repr(e)[:160]
        if world == 1 and not args.consistent and args.with_alternate:
            # informational, outside the timed region: the opt-in alternate-rows mode on the same frame
            alt = ss.passive.StereoASW(winSize=win, maxDisparity=maxD, minDisparity=minD, gammaC=GAMMA_C, gammaP=GAMMA_P,
                                       alternate=True)
            alt_map = alt.compute(ownL, ownR)
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for _ in range(3):
                alt_map = alt.compute(ownL, ownR)
            torch.cuda.synchronize()
            line["alternate_rows_mode"] = {"ms_per_step": (time.perf_counter() - ta) / 3 * 1e3,
                                           "percent_pixels_differing_from_exact": 100.0 * float((alt_map != out).float().mean()),
                                           "note": "opt-in StereoASW(alternate=True); not the reference's output, never `value`"}
        if world == 1 and not args.no_others:
            line["others"] = others(dev, args.seed)
            # the pointwise kernels either side of the matchers (Lab records, rectification remap, 3-D reprojection)
            # against the HBM roofline: tools/bench_pointwise.py
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_pointwise
                line["pointwise_kernels"] = bench_pointwise.measure()
            except Exception as e:      # noqa: BLE001
                line["pointwise_kernels"] = {"error": repr(e)[:200]}
        if world == 1 and not use_dist and not args.no_bad1:
            try:
                line["bad1_vs_cpu_ref"] = bad1_on_reference_strips(dev, consistent=args.consistent, config=args.config)
            except Exception as e:      # noqa: BLE001
                line["bad1_vs_cpu_ref"] = {"percent": None, "source": repr(e)[:200]}
        if world == 1 and not use_dist and not args.no_e2e:
            try:
                resident = {args.config: per_step * 1e3}
                for k in ("default_1080p_d16_w35", "c1_tsukuba_d16_w15"):
                    if k in line.get("others", {}) and "ms_per_step" in line["others"][k]:
                        resident[k] = line["others"][k]["ms_per_step"]
                line["e2e_host_arrays"] = e2e_host_arrays(args.seed, resident)
            except Exception as e:      # noqa: BLE001
                line["e2e_host_arrays"] = {"error": repr(e)[:200]}
        if world == 1 and not use_dist and not args.no_cpu_baseline:
            cb = cpu_baseline(cfg, args.seed, args.cpu_rows_per_thread, crop_cols=args.cpu_crop_cols or cfg[1], repeats=args.cpu_repeats,
                              pin_cores=args.cpu_pin_cores, pinned_rows_per_core=args.cpu_pinned_rows_per_core)
            # second half of BASELINE's metric: % bad-1.0 of the GPU map vs the CPU reference map, on the strip
            # the CPU baseline computed (matched as a stand-alone sub-image by both)
            try:
                ref_map = np.load(cb["map_file"])
                os.remove(cb["map_file"])
                r0s, rws, c0s, cls = cb["strip_row0"], cb["strip_rows"], cb["strip_col0"], cb["strip_cols"]
                gpu_map = matcher.compute(np.ascontiguousarray(L[r0s:r0s + rws, c0s:c0s + cls]),
                                          np.ascontiguousarray(R[r0s:r0s + rws, c0s:c0s + cls]))
                cand, ovf = _native.counter("exact_entries"), _native.counter("exact_overflow")
                diff = np.abs(gpu_map.astype(np.int32) - ref_map.astype(np.int32))
                crop = {"crop_percent": 100.0 * float(np.mean(diff > 1)), "crop_exact_percent": 100.0 * float(np.mean(diff == 0)),
                        "crop_pixels": int(diff.size),
                        "crop_what": "GPU vs CPU %s map of the cpu_baseline crop, a stand-alone %d-column sub-image of a D 0..%d "
                                     "frame: %d of its columns have truncated candidate sets whose costs saturate, so ties between "
                                     "equal costs are frequent (see the tie breakdown); the headline `percent` is taken on "
                                     "full-width strips" % (cb["kind"], cls, maxD, min(cls, maxD))}
                line.setdefault("bad1_vs_cpu_ref", {}).update(crop)
                # how many of the differing pixels are numerical ties: the raw GPU cost at the reference's disparity is
                # within the stated raw-cost tolerance (1e-4 relative) of the GPU's own minimum -- either choice is a
                # minimum within the arithmetic (a stand-alone 512-column crop of a D 0..192 frame has wide bands where most
                # candidates fall outside the image and the truncated costs saturate)
                try:
                    cl = np.ascontiguousarray(L[r0s:r0s + rws, c0s:c0s + cls])
                    cr = np.ascontiguousarray(R[r0s:r0s + rws, c0s:c0s + cls])
                    costs = np.empty((rws, cls, nD), np.float32)
                    _native.check(lib.ssamd_asw_costs(cl.ctypes.data, cr.ctypes.data, rws, cls, win, maxD, minD, float(GAMMA_C), float(GAMMA_P),
                                                      costs.ctypes.data, -1))
                    yy, xx = np.mgrid[0:rws, 0:cls]
                    gi = np.clip(gpu_map.astype(np.int64) - minD, 0, nD - 1)
                    ri = np.clip(ref_map.astype(np.int64) - minD, 0, nD - 1)
                    cg, cr_ = costs[yy, xx, gi], costs[yy, xx, ri]
                    tie = np.abs(cr_ - cg) <= 1e-4 * np.maximum(1.0, np.abs(cg))
                    line["bad1_vs_cpu_ref"]["crop_percent_excluding_numerical_ties"] = 100.0 * float(np.mean((diff > 1) & ~tie))
                    line["bad1_vs_cpu_ref"]["crop_numerical_ties_among_bad1_percent"] = 100.0 * float(np.mean((diff > 1) & tie))
                    tie6 = np.abs(cr_ - cg) <= 1e-6 * np.maximum(1.0, np.abs(cg))
                    line["bad1_vs_cpu_ref"]["crop_percent_excluding_ties_at_1e-6"] = 100.0 * float(np.mean((diff > 1) & ~tie6))
                    line["bad1_vs_cpu_ref"]["crop_exact_mode"] = {"percent": crop["crop_percent"], "exact_percent": crop["crop_exact_percent"],
                                                                  "candidates_reevaluated": cand, "queue_overflow": ovf}
                    # the same crop WITHOUT the fp64 tie-break pass (exact=False, the default of rounds 1-5): saturated candidates whose
                    # fp64 costs agree to the last ulps are where the fp32 argmin and the reference part
                    xm = ss.passive.StereoASW(winSize=win, maxDisparity=maxD, minDisparity=minD, gammaC=GAMMA_C, gammaP=GAMMA_P,
                                              consistent=bool(args.consistent), exact=False).compute(cl, cr)
                    xdiff = np.abs(xm.astype(np.int32) - ref_map.astype(np.int32))
                    line["bad1_vs_cpu_ref"]["crop_fp32_mode"] = {"percent": 100.0 * float(np.mean(xdiff > 1)),
                                                                 "exact_percent": 100.0 * float(np.mean(xdiff == 0)),
                                                                 "pixels_changed_by_the_tie_break": int(np.count_nonzero(xm != gpu_map))}
                except Exception as e:      # noqa: BLE001
                    line["bad1_vs_cpu_ref"]["crop_percent_excluding_numerical_ties"] = repr(e)[:120]
            except Exception as e:      # noqa: BLE001
                line.setdefault("bad1_vs_cpu_ref", {})["crop_what"] = repr(e)[:160]
            cb.pop("map_file", None)
            line["cpu_baseline"] = cb
            if cb["value"]:
                line["speedup_vs_cpu_baseline"] = line["value"] / cb["value"]
                line["speedup_vs_cpu_baseline_range"] = [line["value"] / cb["value_max"], line["value"] / cb["value_min"]]
                line["speedup_vs_cpu_baseline_cores"] = cb["cores"]
                # GPU against ONE busy host core (effective cores = process CPU seconds / wall seconds of the sample):
                # divide by a core count to get the ratio against that many cores (north_star's 2000 x holds up to
                # speedup_per_effective_core / 2000 cores of this host)
                if cb.get("value_per_effective_core"):
                    line["speedup_per_effective_core"] = line["value"] / cb["value_per_effective_core"]
                    line["cores_at_which_speedup_is_2000x"] = line["speedup_per_effective_core"] / 2000.0
                if cb.get("all_threads", {}).get("value"):
                    line["speedup_vs_cpu_all_host_threads"] = line["value"] / cb["all_threads"]["value"]
                if cb.get("hoisted", {}).get("value"):
                    line["speedup_vs_cpu_hoisted_port"] = line["value"] / cb["hoisted"]["value"]
        result = emit(line)
    else:
        result = None
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()      # RCCL may print banner lines on teardown: keep the JSON line last
    sys.stdout.flush()
    if result is not None:
        os.write(real_stdout, (result + "\n").encode())


if __name__ == "__main__":
    main()
