"""Row-strip partitioning of one stereo pair across the GPUs of a node.

Output row y of ASW/GSW depends only on input rows y-pad .. y+pad of both images
(pad = winSize // 2; reference ``_passive.cpp:38-40, 60-62``) and the left-right
check / occlusion fill only touches row y (``:251-285``); the reference already
treats rows as independent jobs (``:372-374``).  So the image is cut into
contiguous strips of rows, one per rank (one process per GPU); each rank needs
``pad`` extra *input* rows above and below its strip -- the halo -- which it
receives from the ranks owning them, point-to-point over RCCL/xGMI
(``torch.distributed`` backend ``nccl``; ``gloo`` on CPU for the tests).  No
intermediate data is exchanged: a strip that carries its full halo reproduces
the whole-image rows bit-exactly, because the kernels treat the sub-image border
as the image border only where it *is* the image border.

    own   = rows [r0, r1) of the image         (resident on this rank)
    sub   = rows [h0, h1) = own + halos         (after exchange_halos)
    out   = matcher rows [r0-h0, r1-h0) of sub  (strip of the disparity map)

The halo exchange is one batched group of isend/irecv (ncclGroupStart/End under
the hood), at most a few messages of pad*W*3 bytes per neighbour and image; the
strips of the result are collected with one all_gather on equal-padded strips.
"""
import numpy as np

__all__ = ["strip_bounds", "halo_bounds", "transfer_plan", "exchange_halos", "gather_strips", "match_strip", "matcher_pad",
           "StripContext", "p2p_self_probe"]


def strip_bounds(height, world_size, rank):
    """Rows [r0, r1) owned by `rank`: contiguous, sizes differ by at most one row."""
    base, extra = divmod(int(height), int(world_size))
    r0 = rank * base + min(rank, extra)
    r1 = r0 + base + (1 if rank < extra else 0)
    return r0, r1


def halo_bounds(height, r0, r1, pad):
    """Input rows [h0, h1) needed to compute output rows [r0, r1)."""
    if r1 <= r0:
        return r0, r0
    return max(0, r0 - pad), min(int(height), r1 + pad)


def transfer_plan(height, world_size, pad):
    """All point-to-point messages of one halo exchange.

    Returns a list of ``(src_rank, dst_rank, row_begin, row_end)``: global rows
    [row_begin,row_end) owned by src that dst needs as halo.  Deterministic and
    identical on every rank, so sends and receives pair up without negotiation.
    Handles strips thinner than the halo (rows then come from several ranks).
    """
    own = [strip_bounds(height, world_size, r) for r in range(world_size)]
    plan = []
    for dst in range(world_size):
        r0, r1 = own[dst]
        h0, h1 = halo_bounds(height, r0, r1, pad)
        for src in range(world_size):
            if src == dst:
                continue
            s0, s1 = own[src]
            for lo, hi in ((max(h0, s0), min(r0, s1)), (max(r1, s0), min(h1, s1))):
                if hi > lo:
                    plan.append((src, dst, lo, hi))
    return plan


def exchange_halos(own_left, own_right, height, pad, rank, world_size, group=None):
    """Assemble this rank's sub-image (strip + halo rows) of both images.

    own_left / own_right: torch uint8 tensors [r1-r0, W, 3] holding the rows this
    rank owns (CUDA tensors with the nccl backend, CPU tensors with gloo).
    Returns ``(sub_left, sub_right, out_row0, out_rows)``.
    """
    import torch
    import torch.distributed as dist
    r0, r1 = strip_bounds(height, world_size, rank)
    h0, h1 = halo_bounds(height, r0, r1, pad)
    W = own_left.shape[1]
    assert own_left.shape[0] == r1 - r0 and own_right.shape == own_left.shape
    subs = []
    for own in (own_left, own_right):
        sub = torch.empty((h1 - h0, W, 3), dtype=own.dtype, device=own.device)
        sub[r0 - h0:r1 - h0] = own
        subs.append(sub)
    if world_size > 1:
        ops, keep = [], []
        for src, dst, lo, hi in transfer_plan(height, world_size, pad):
            for own, sub in zip((own_left, own_right), subs):
                if src == rank:
                    buf = own[lo - r0:hi - r0].contiguous()
                    keep.append(buf)
                    ops.append(dist.P2POp(dist.isend, buf, dst, group=group))
                elif dst == rank:
                    ops.append(dist.P2POp(dist.irecv, sub[lo - h0:hi - h0], src, group=group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
    return subs[0], subs[1], r0 - h0, r1 - r0


def gather_strips(strip_disparity, height, rank, world_size, group=None):
    """All-gather the per-rank disparity strips into the full [height, W] map (on every rank)."""
    import torch
    import torch.distributed as dist
    if world_size == 1 and not dist.is_initialized():
        return strip_disparity
    W = strip_disparity.shape[1]
    rows_max = -(-int(height) // world_size)
    padded = torch.zeros((rows_max, W), dtype=strip_disparity.dtype, device=strip_disparity.device)
    padded[:strip_disparity.shape[0]] = strip_disparity
    # int16 is not a NCCL/RCCL (nor gloo) collective dtype: move the strips as raw bytes
    wire = padded.view(torch.uint8)
    parts = [torch.empty_like(wire) for _ in range(world_size)]
    dist.all_gather(parts, wire, group=group)
    out = torch.empty((int(height), W), dtype=strip_disparity.dtype, device=strip_disparity.device)
    for r in range(world_size):
        r0, r1 = strip_bounds(height, world_size, r)
        out[r0:r1] = parts[r].view(strip_disparity.dtype)[:r1 - r0]
    return out


def matcher_pad(matcher):
    """Halo rows a strip of this matcher needs: winSize // 2, and one more for ``StereoASW(alternate=True)``, whose
    odd rows take their candidates from the exactly matched rows above and below."""
    return int(matcher.winSize) // 2 + (1 if getattr(matcher, "alternate", False) else 0)


def match_strip(matcher, own_left, own_right, height, rank, world_size, group=None, gather=True):
    """One distributed `compute`: halo exchange -> kernels on the strip -> (optional) gather.

    matcher: a ``StereoASW`` / ``StereoGSW`` instance; the tensors must be device
    tensors (the kernels run on the tensors' GPU on the current stream).
    """
    pad = matcher_pad(matcher)
    subL, subR, out_row0, out_rows = exchange_halos(own_left, own_right, height, pad, rank, world_size, group)
    r0, _ = strip_bounds(height, world_size, rank)
    strip = _match_rows(matcher, subL, subR, out_row0, out_rows, (r0 - out_row0) & 1)
    return gather_strips(strip, height, rank, world_size, group) if gather else strip


def _match_rows(matcher, subL, subR, out_row0, out_rows, row_parity=0):
    """The kernels on one strip.  A rank whose strip is EMPTY (more ranks than image rows) has nothing to match --
    the operators refuse a 0-row image -- but must still take part in the collectives that follow, so it returns an
    empty int16 strip instead of raising while its peers wait in the all_gather."""
    import torch
    if out_rows <= 0:
        return torch.empty((0, int(subL.shape[1])), dtype=torch.int16, device=subL.device)
    if getattr(matcher, "alternate", False):          # row_parity: parity of the sub-image's first row in the whole image
        return matcher._compute_device(subL, subR, out_row0=out_row0, out_rows=out_rows, row_parity=row_parity)
    return matcher._compute_device(subL, subR, out_row0=out_row0, out_rows=out_rows)


class StripContext:
    """Reusable state of one rank for repeated frames of the same geometry: the transfer plan,
    the sub-image buffers (strip + halos) and the gather buffers are built once, so that a step is
    two device copies, one batched isend/irecv group, the kernels and one all_gather_into_tensor.

        ctx = StripContext(matcher, height, width, rank, world_size, device)
        full = ctx.step(own_left, own_right)        # full [height, width] int16 map on every rank

    With the nccl (RCCL) backend all messages move device to device.  With gloo and a GPU `device`
    (functional tests of the multi-process flow on a box without RCCL peers) the messages are staged
    through host buffers; the kernels still run on `device`.
    """

    def __init__(self, matcher, height, width, rank, world_size, device, group=None, overlap=True, loopback=False):
        """loopback=True (TIMING HARNESS, tools/strip_host_cost.py): `rank` / `world_size` are VIRTUAL -- the strip, halo and message
        sizes of that rank of that world -- while every message goes to this very process (isend / irecv to self in one batched
        group, the p2p_self_probe pattern) and nothing is gathered: the real step of an 8-GPU run (copies, RCCL group on the side
        stream, interior launch, border launch) with its real sizes on a box with ONE GPU.  The halo rows it receives are its own
        border rows, so the map is NOT the whole frame's: for timing only."""
        import torch
        import torch.distributed as dist
        self.matcher, self.H, self.W = matcher, int(height), int(width)
        self.rank, self.world, self.group = int(rank), int(world_size), group
        self.loopback = bool(loopback)
        self.device = torch.device(device)
        backend = dist.get_backend(group) if dist.is_initialized() else None
        self.flat_gather = backend == "nccl"                  # gloo has no all_gather_into_tensor
        self.staged = backend == "gloo" and self.device.type != "cpu"
        cdev = torch.device("cpu") if self.staged else self.device          # where messages live
        self.pad = matcher_pad(matcher)
        self.r0, self.r1 = strip_bounds(self.H, self.world, self.rank)
        self.h0, self.h1 = halo_bounds(self.H, self.r0, self.r1, self.pad)
        self.subL = torch.zeros((self.h1 - self.h0, self.W, 3), dtype=torch.uint8, device=self.device)
        self.subR = torch.zeros_like(self.subL)
        mine = [t for t in transfer_plan(self.H, self.world, self.pad) if self.rank in (t[0], t[1])]
        self.sends = [(dst, lo - self.r0, hi - self.r0) for src, dst, lo, hi in mine if src == self.rank]
        self.recvs = [(src, lo - self.h0, hi - self.h0) for src, dst, lo, hi in mine if dst == self.rank]

        def pair(n):
            return (torch.empty((n, self.W, 3), dtype=torch.uint8, device=cdev),
                    torch.empty((n, self.W, 3), dtype=torch.uint8, device=cdev))
        self.send_bufs = [pair(hi - lo) for _, lo, hi in self.sends]
        self.recv_bufs = [pair(hi - lo) for _, lo, hi in self.recvs] if self.staged else None
        # the message list of a step never changes (fixed buffers, fixed peers): built once
        self._ops = []
        if self.world > 1 and dist.is_initialized():
            me = dist.get_rank(self.group) if self.loopback else None
            for (dst, lo, hi), (bl, br) in zip(self.sends, self.send_bufs):
                self._ops.append(dist.P2POp(dist.isend, bl, me if self.loopback else dst, group=self.group))
                self._ops.append(dist.P2POp(dist.isend, br, me if self.loopback else dst, group=self.group))
            for k, (src, lo, hi) in enumerate(self.recvs):
                tl, tr = self.recv_bufs[k] if self.staged else (self.subL[lo:hi], self.subR[lo:hi])
                self._ops.append(dist.P2POp(dist.irecv, tl, me if self.loopback else src, group=self.group))
                self._ops.append(dist.P2POp(dist.irecv, tr, me if self.loopback else src, group=self.group))
        self.rows_max = -(-self.H // self.world)
        self.padded = torch.zeros((self.rows_max, self.W), dtype=torch.int16, device=cdev)
        self.gathered = torch.empty((self.world, self.rows_max, self.W), dtype=torch.int16, device=cdev)
        self.full = torch.empty((self.H, self.W), dtype=torch.int16, device=self.device)
        self.backend = backend
        # Overlap (round 5): the rows whose windows stay inside the rows this rank OWNS do not need the halo -- they are matched
        # while the halo rows are in flight on a side stream; the (at most 2 * pad) border rows follow as ONE launch
        # (ssamd_asw_device_rows2).  Rows are independent jobs (_passive.cpp:372-374): the strip's map does not depend on the cut.
        n = self.r1 - self.r0
        self.top = min(self.pad, n) if self.h0 < self.r0 else 0        # rows that read halo rows above / below
        self.bot = min(self.pad, n - self.top) if self.h1 > self.r1 else 0
        self.interior = n - self.top - self.bot
        # Two launches instead of one must not cost more rounds of workgroups than they hide exchange time: the interior launch is
        # cut to whole rounds (8 GPUs, config 3: 135-row strip = 101 interior rows = 6.3 rounds of 256 workgroups at 16 per row ->
        # 96 rows = 6 rounds; the other 39 rows = 2.4 rounds go to the border launch, whose last part-filled round runs as
        # half-width tiles), so that interior + border = the rounds of one launch over the strip
        import os
        mode = os.environ.get("SSAMD_STRIP_OVERLAP", "1")             # "0": the sequential step of rounds 1-4; "force": overlap wherever
        if mode != "force":                                            # a row is free of the halo, whatever the rounds (tests)
            self.interior = self._whole_rounds(matcher, self.interior)
        self.bot = n - self.top - self.interior
        if mode == "0":
            overlap = False
        # (round 6: StereoGSW too -- ssamd_gsw_device_rows2; exact mode rides along -- ssamd_asw_exact_device_rows2)
        # StereoGSW has the two-range form as well, but its step is NOT overlapped by default: measured (tools/strip_host_cost.py,
        # profiles/r06_strip_host_cost.txt) the 135-row strip of an 8-GPU config-4 run takes 1.28 ms sequentially and 1.48 ms
        # overlapped -- the two 5-row border bands cost more launches than the 0.13 ms exchange they hide.  "force" turns it on.
        # Likewise StereoASW(consistent=True): 5.73-5.96 ms overlapped against 5.51-5.58 ms sequentially for the same strip -- there the
        # RCCL kernels only got onto the GPU when the interior rows' aggregation kernel had drained (TORCH_NCCL_HIGH_PRIORITY or not),
        # so the exchange was exposed anyway and the second launch was pure cost.  Plain StereoASW: 5.40 vs 5.39 ms with the exchange
        # fully hidden (exposed 0.0 ms) -- the default there, because on real xGMI peers the exchange can only be slower than the
        # 0.12 ms of this loopback.  "force" / "all": overlap every matcher that has a two-range form.
        everything = mode in ("force", "all", "gsw")
        kinds = ("StereoASW", "StereoGSW") if everything else ("StereoASW",)
        if type(matcher).__name__ == "StereoASW" and getattr(matcher, "consistent", False) and not everything:
            overlap = False
        self.overlap = bool(overlap and self.device.type == "cuda" and type(matcher).__name__ in kinds and
                            not getattr(matcher, "alternate", False) and
                            self.world > 1 and self._ops and self.interior > 0 and self.top + self.bot > 0)
        self.side = torch.cuda.Stream(device=self.device) if self.overlap else None
        self.strip_out = torch.empty((n, self.W), dtype=torch.int16, device=self.device) if self.overlap else None
        # the two cross-stream events of a step are made ONCE (a torch.cuda.Event() per step was two allocations on the host path)
        self._ev_copied = torch.cuda.Event() if self.overlap else None
        self._ev_arrived = torch.cuda.Event() if self.overlap else None
        self.host_s, self.host_steps = 0.0, 0          # host time spent inside step() (no synchronisation): read_timing()["host_step_ms"]
        self._timing = None           # list of per-step event tuples while enable_timing() is on (GPU devices only)

    def _whole_rounds(self, matcher, interior):
        """the largest number of rows <= interior whose workgroups fill whole rounds of the device (0 when that is less than one
        round or the geometry is unknown: a strip that small is not worth two launches)"""
        if interior <= 0 or self.device.type != "cuda" or type(matcher).__name__ != "StereoASW":
            return max(interior, 0)
        try:
            import torch
            from . import _native
            g = _native.asw_geometry(self.W, interior, int(matcher.winSize), int(matcher.maxDisparity), int(matcher.minDisparity))
            if _native.asw_kernel_form(self.W, interior, int(matcher.winSize), int(matcher.maxDisparity), int(matcher.minDisparity))["wave_kernel"]:
                return interior                      # wave kernels: thousands of independent waves, no round structure to respect
            per_row = g["grid_x"] * g["grid_z"]
            waves = g["threads"] // 64
            resident = max(1, min(3 // max(1, (waves + 3) // 4), (160 * 1024) // max(1, g["lds_bytes"])))
            slots = torch.cuda.get_device_properties(self.device).multi_processor_count * resident
            rounds = interior * per_row // slots
            return int(rounds * slots // per_row) if rounds >= 1 else 0
        except Exception:      # noqa: BLE001 -- a heuristic: never in the way of the step
            return interior

    def enable_timing(self, on=True):
        """Record device events around the phases of every step (halo exchange incl. the strip copies, kernels, gather) on
        the streams they run on; read_timing() returns their per-step averages."""
        self._timing = [] if on and self.device.type == "cuda" else None

    def read_timing(self):
        """{'steps', 'exchange_ms', 'kernels_ms', 'gather_ms'} per step since enable_timing(); with the overlapped step also
        'interior_ms' (kernels that ran while the halo was in flight), 'border_ms' and 'exchange_exposed_ms' = the part of
        the exchange that was NOT hidden under the interior rows (0 when it ended first)."""
        import torch
        if not self._timing:
            return None
        torch.cuda.synchronize(self.device)
        n = len(self._timing)
        out = {"steps": n, "overlapped": bool(self._timing[0].get("x0") is not None)}
        if out["overlapped"]:
            out["exchange_ms"] = sum(t["x0"].elapsed_time(t["x1"]) for t in self._timing) / n
            out["interior_ms"] = sum(t["e1"].elapsed_time(t["ei"]) for t in self._timing) / n
            out["border_ms"] = sum(t["ei"].elapsed_time(t["e2"]) for t in self._timing) / n
            out["exchange_exposed_ms"] = sum(max(0.0, t["ei"].elapsed_time(t["x1"])) for t in self._timing) / n
            out["copies_ms"] = sum(t["e0"].elapsed_time(t["e1"]) for t in self._timing) / n
        else:
            out["exchange_ms"] = sum(t["e0"].elapsed_time(t["e1"]) for t in self._timing) / n
        out["kernels_ms"] = sum(t["e1"].elapsed_time(t["e2"]) for t in self._timing) / n
        out["gather_ms"] = sum(t["e2"].elapsed_time(t["e3"]) for t in self._timing) / n
        # host side of a step: time the CPU spends inside step() enqueueing it (copies, event records, the RCCL group, two matcher
        # calls, the gather) -- must stay well below the strip's kernel time or the GPU waits for its host (a 135-row strip of an
        # 8-GPU config-3 run is ~5 ms); measured with the timing events on, i.e. an upper bound
        out["host_step_ms"] = sum(t["host_s"] for t in self._timing) / n * 1e3
        return out

    def _mark(self, stream=None):
        import torch
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream if stream is not None else torch.cuda.current_stream(self.device))
        return e

    def _exchange(self, own_left, own_right):
        """the halo messages of one step on the CURRENT stream: fill the send buffers, one batched isend / irecv group, unpack"""
        import torch.distributed as dist
        for (dst, lo, hi), (bl, br) in zip(self.sends, self.send_bufs):
            bl.copy_(own_left[lo:hi])
            br.copy_(own_right[lo:hi])
        for req in dist.batch_isend_irecv(self._ops):
            req.wait()
        if self.staged:
            for (src, lo, hi), (tl, tr) in zip(self.recvs, self.recv_bufs):
                self.subL[lo:hi].copy_(tl)
                self.subR[lo:hi].copy_(tr)

    def step(self, own_left, own_right, gather=True):
        """One frame.  NOTE (aliasing): with the overlapped step the strip is written into the context's persistent buffer
        `strip_out` -- with gather=False the caller gets THAT buffer, which the next step() overwrites asynchronously on the
        stream; clone it to keep it.  With gather=True the result is the context's gather buffer, likewise reused."""
        import time
        import torch
        host0 = time.perf_counter()
        timing = self._timing is not None
        t = {"e0": self._mark()} if timing else None
        o0, o1 = self.r0 - self.h0, self.r1 - self.h0
        self.subL[o0:o1].copy_(own_left)
        self.subR[o0:o1].copy_(own_right)
        if self.overlap:
            main = torch.cuda.current_stream(self.device)
            copied, arrived = self._ev_copied, self._ev_arrived
            copied.record(main)
            if timing:
                t["e1"] = self._mark()
            # interior rows first (asynchronous launches on the main stream) ...
            self.matcher._compute_device(self.subL, self.subR, out_row0=o0 + self.top, out_rows=self.interior,
                                         out=self.strip_out[self.top:self.top + self.interior])
            if timing:
                t["ei"] = self._mark()
            # ... the halo meanwhile on the side stream (RCCL orders its work behind the stream that is current at the call)
            with torch.cuda.stream(self.side):
                self.side.wait_event(copied)
                if timing:
                    t["x0"] = self._mark(self.side)
                self._exchange(own_left, own_right)
                if timing:
                    t["x1"] = self._mark(self.side)
                arrived.record(self.side)
            main.wait_event(arrived)
            # ... then the border bands, one launch
            strip = self.matcher._compute_device(self.subL, self.subR, out_row0=o0, out_rows=self.r1 - self.r0, out=self.strip_out,
                                                 skip=(o0 + self.top, self.interior))
        else:
            if self._ops:
                self._exchange(own_left, own_right)
            if timing:
                t["e1"] = self._mark()
            strip = _match_rows(self.matcher, self.subL, self.subR, o0, self.r1 - self.r0, self.h0 & 1)
        if timing:
            t["e2"] = self._mark()
        out = self._gather(strip) if gather and not self.loopback else strip
        dt = time.perf_counter() - host0
        self.host_s += dt
        self.host_steps += 1
        if timing:
            t["e3"] = self._mark()
            t["host_s"] = dt
            self._timing.append(t)
        return out

    def _gather(self, strip):
        import torch
        import torch.distributed as dist
        if self.world == 1 and not dist.is_initialized():
            return strip
        self.padded[:strip.shape[0]].copy_(strip)
        # int16 is not an RCCL collective dtype: gather the strips as bytes
        if self.flat_gather:
            dist.all_gather_into_tensor(self.gathered.view(torch.uint8), self.padded.view(torch.uint8), group=self.group)
        else:
            parts = list(self.gathered.view(torch.uint8).unbind(0))
            dist.all_gather(parts, self.padded.view(torch.uint8), group=self.group)
        if self.H == self.rows_max * self.world and not self.staged:
            return self.gathered.view(self.H, self.W)
        for r in range(self.world):
            a, b = strip_bounds(self.H, self.world, r)
            self.full[a:b].copy_(self.gathered[r, :b - a])
        return self.full


def p2p_self_probe(device, nbytes=98304, group=None):
    """One batched isend/irecv group in which this rank is its own peer (ncclSend / ncclRecv to self inside one
    group): the device-to-device point-to-point path of the halo exchange, exercisable on a box with ONE GPU, where
    no second RCCL rank can exist.  Returns True when the bytes arrive.  Needs an initialised process group."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    src = torch.arange(nbytes, dtype=torch.int64, device=device).to(torch.uint8)
    dst = torch.zeros_like(src)
    ops = [dist.P2POp(dist.isend, src, rank, group=group), dist.P2POp(dist.irecv, dst, rank, group=group)]
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    return bool(torch.equal(src, dst))


def split_rows(image, world_size):
    """Host-side helper: the list of row strips of a numpy image (views)."""
    H = image.shape[0]
    return [np.ascontiguousarray(image[slice(*strip_bounds(H, world_size, r))]) for r in range(world_size)]
