"""GPU tier: the multi-process strip flow with the REAL kernels.  A 1-GPU box cannot host two RCCL ranks
(RCCL refuses duplicate devices), so the ranks share GPU 0 for the kernels and exchange halos / gather strips
over gloo through host staging (StripContext's staged mode); the message plan, the sub-image calls
(out_row0 / out_rows) and the reassembly are the ones the RCCL run uses."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _has_golden(cid):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full_cases.npz")
    try:
        return cid in np.load(path).files
    except OSError:
        return False


def _worker(rank, world, port, case, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if case[3].get("_overlap") is not None:
        os.environ["SSAMD_STRIP_OVERLAP"] = "force"     # small test frames: overlap wherever a row is free of the halo, whatever the rounds
    try:
        import simplestereo_amd as ss
        from simplestereo_amd import strips
        from simplestereo_amd.synth import make_pair
        algo, H, W, params = case
        L, R, _ = make_pair(H, W, params["maxDisparity"], params.get("_seed", 11))
        dev = torch.device("cuda", 0)
        m = (ss.passive.StereoASW if algo == "asw" else ss.passive.StereoGSW)(**{k: v for k, v in params.items() if not k.startswith("_")})
        r0, r1 = strips.strip_bounds(H, world, rank)
        ownL = torch.from_numpy(np.ascontiguousarray(L[r0:r1])).to(dev)
        ownR = torch.from_numpy(np.ascontiguousarray(R[r0:r1])).to(dev)
        ctx = strips.StripContext(m, H, W, rank, world, dev)
        assert ctx.staged
        full = None
        for _ in range(2):                                   # the context is reusable across frames
            full = ctx.step(ownL, ownR).cpu().numpy().copy()
        if params.get("_golden"):
            # the reference's own map of the whole frame (tests/golden/full_cases.npz, made by the unmodified _passive.cpp)
            want = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full_cases.npz"))[params["_golden"]]
        else:
            want = m.compute(L, R)                           # whole frame, one process
        exp = params.get("_overlap")
        exp = exp[rank] if isinstance(exp, (list, tuple)) else exp
        q.put((rank, bool(np.array_equal(full, want)) and (exp is None or ctx.overlap == exp),
               int((full != want).sum()) if exp is None or ctx.overlap == exp else "overlap=%r" % ctx.overlap))
    except Exception as e:      # noqa: BLE001
        q.put((rank, False, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [
    (2, ("asw", 61, 200, dict(winSize=21, maxDisparity=40, consistent=True, _overlap=True))),
    (3, ("asw", 50, 160, dict(winSize=35, maxDisparity=24, minDisparity=2))),      # strips thinner than the halo
    (2, ("gsw", 45, 150, dict(winSize=11, maxDisparity=30))),
    # round 6: StereoGSW rides the overlapped step too (ssamd_gsw_device_rows2), and so does exact mode, which is what StereoASW() runs by
    # default (ssamd_asw_exact_device_rows2): every ASW case below goes through the fp64 tie-break pass on each rank; one without it
    (3, ("gsw", 70, 180, dict(winSize=11, maxDisparity=40, _overlap=True))),
    (2, ("asw", 90, 200, dict(winSize=15, maxDisparity=40, exact=False, _overlap=True))),
    # round 5, the overlapped step: interior rows matched while the halo is in flight, the border bands as one launch --
    # direct-write path (one disparity chunk, no right pass) and keyed path; a middle rank with two bands; the wave kernel
    (2, ("asw", 90, 200, dict(winSize=15, maxDisparity=40, _overlap=True))),
    (3, ("asw", 120, 260, dict(winSize=15, maxDisparity=70, consistent=True, _overlap=True))),
    # 32-row strips, 17-row halo: the middle rank has no row that is free of both halos, the outer ranks have 15 (one halo each)
    (3, ("asw", 96, 300, dict(winSize=35, maxDisparity=16, minDisparity=1, _overlap=[True, False, True]))),
    (2, ("asw", 100, 300, dict(winSize=35, maxDisparity=16, _overlap=True))),
    # default mode (interior launch cut to whole rounds of workgroups, the rest joins the border launch incl. its half-width last round)
    (2, ("asw", 200, 1920, dict(winSize=35, maxDisparity=192))),
    (3, ("asw", 47, 160, dict(winSize=11, maxDisparity=24, alternate=True))),      # strips of 16 / 16 / 15 rows: odd and even starts
    # BASELINE config 5's partition: eight ranks, 4096 columns, D 0..256, win 35 (the 88 x 260 tiles of the 4K launch); 136
    # rows = eight strips of 17 rows = exactly the halo, so every interior rank receives both halos whole from its neighbours
    (8, ("asw", 136, 4096, dict(winSize=35, maxDisparity=256))),
    # round 6: BASELINE config 5 ITSELF -- the whole 4096 x 2160 frame, D 0..256, win 35, cut into eight strips, against the map the
    # unmodified reference computed for it (F5p, 1.6 h of its time): every pixel equal
    pytest.param(8, ("asw", 2160, 4096, dict(winSize=35, maxDisparity=256, _seed=1, _golden="F5p")),
                 marks=pytest.mark.skipif(not _has_golden("F5p"), reason="F5p not generated (tests/golden/make_golden_full.py F5p)")),
])
def test_strips_across_processes_reproduce_the_whole_frame(world, case):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=480) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
