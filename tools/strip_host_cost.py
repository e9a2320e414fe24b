"""GPU box (round 6): the step of ONE rank of an 8-GPU (or 4-, 2-GPU) run, with its real sizes, on a box with one GPU.

StripContext(loopback=True) takes the strip / halo / message geometry of a VIRTUAL rank of a virtual world while every message
goes to this very process over RCCL (isend / irecv to self in one batched group): copies, the RCCL group on the side stream,
the interior launch, the border launch -- everything but the all_gather, which needs peers.  Reported per configuration:
  wall_ms        per step, device-synchronised once after the loop
  host_step_ms   CPU time inside step() per step (no synchronisation) -- the budget VERDICT r05 set is <= 0.25 ms
  kernel_ms      aggregation kernels per step (interior + border launch), exchange_ms, exchange_exposed_ms (not hidden)
Not a scaling measurement: one GPU, the peer is the rank itself."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")      # (as bench.py sets it; TORCH_NCCL_HIGH_PRIORITY=0 in the environment for the A/B)
import numpy as np, torch, torch.distributed as dist
import simplestereo_amd as ss
from simplestereo_amd import _native, strips
from simplestereo_amd.synth import make_pair

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
lib = _native.lib()
H, W, maxD = 1080, 1920, 192
L, R, _ = make_pair(H, W, maxD, 1)
cases = [("asw c3 exact (default)", ss.passive.StereoASW(winSize=35, maxDisparity=maxD), _native.K_ASW_AGG),
         ("asw c3 fp32", ss.passive.StereoASW(winSize=35, maxDisparity=maxD, exact=False), _native.K_ASW_AGG),
         ("asw c3 consistent", ss.passive.StereoASW(winSize=35, maxDisparity=maxD, consistent=True), _native.K_ASW_AGG),
         ("gsw c4", ss.passive.StereoGSW(winSize=11, maxDisparity=maxD), _native.K_GSW_AGG)]
print(json.dumps({"TORCH_NCCL_HIGH_PRIORITY": os.environ["TORCH_NCCL_HIGH_PRIORITY"]}), flush=True)
WORLDS = ((8, 3), (4, 1), (2, 0)) if len(sys.argv) < 2 else ((int(sys.argv[1]), int(sys.argv[2])),)
for world, vrank in WORLDS:
    r0, r1 = strips.strip_bounds(H, world, vrank)
    ownL = torch.from_numpy(np.ascontiguousarray(L[r0:r1])).to(dev)
    ownR = torch.from_numpy(np.ascontiguousarray(R[r0:r1])).to(dev)
    for name, m, slot in cases:
        for overlap in (True, False):
            os.environ["SSAMD_STRIP_OVERLAP"] = "gsw" if overlap else "0"      # ("gsw": the default cut + StereoGSW, whose overlap is off by default)
            ctx = strips.StripContext(m, H, W, vrank, world, dev, loopback=True)
            for _ in range(3):
                ctx.step(ownL, ownR, gather=False)
            torch.cuda.synchronize()
            # (a) without any timing events: wall and host
            n = 30
            ctx.host_s, ctx.host_steps = 0.0, 0
            t0 = time.perf_counter()
            for _ in range(n):
                ctx.step(ownL, ownR, gather=False)
            host_loop = time.perf_counter() - t0
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / n * 1e3
            host = ctx.host_s / ctx.host_steps * 1e3
            # (b) with the context's events and the library's kernel events
            ctx.enable_timing(True)
            lib.ssamd_profile_enable(1); lib.ssamd_profile_reset()
            for _ in range(10):
                ctx.step(ownL, ownR, gather=False)
            torch.cuda.synchronize()
            ms, launches = _native.profile_read(); lib.ssamd_profile_enable(0)
            ph = ctx.read_timing(); ctx.enable_timing(False)
            print(json.dumps({"case": name, "virtual_rank": "%d of %d" % (vrank, world), "strip_rows": r1 - r0, "overlapped": ctx.overlap,
                              "interior_rows": ctx.interior if ctx.overlap else None, "border_rows": (ctx.top + ctx.bot) if ctx.overlap else None,
                              "wall_ms": round(wall, 3), "host_step_ms": round(host, 3), "host_enqueue_of_%d_steps_ms" % n: round(host_loop * 1e3, 2),
                              "kernel_ms": round(ms[slot] / 10, 3), "exact_pass_ms": round(ms[_native.K_ASW_EXACT] / 10, 3),
                              "exchange_ms": round(ph.get("exchange_ms", 0.0), 3),
                              "exchange_exposed_ms": round(ph["exchange_exposed_ms"], 3) if "exchange_exposed_ms" in ph else None,
                              "host_step_ms_with_events": round(ph["host_step_ms"], 3)}), flush=True)
dist.destroy_process_group()
