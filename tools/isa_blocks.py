#!/usr/bin/env python3
"""Per-basic-block instruction classes of one kernel in a hipcc -S listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only -o ssamd.s simplestereo_amd/csrc/ssamd_api.hip
    python tools/isa_blocks.py ssamd.s asw_aggregate_pipe_kernelILb0 [min_fma]

Prints, for every basic block with at least `min_fma` v_fma_f32, the count of each instruction class (the tap
instructions v_mul_f32 / v_fma_f32, e unpack v_cvt_f32_ubyte* / v_sub_f32, address v_add*, moves, lane accesses,
LDS, scalar, waits) -- the evidence behind DESIGN.md's issued/useful figures."""
import collections
import re
import sys


def classify(op):
    if op in ("v_fma_f32", "v_fmac_f32"):
        return "tap:v_fma_f32"
    if op == "v_mul_f32":
        return "tap:v_mul_f32"
    if op.startswith("v_pk_"):
        return "valu:" + op
    if op.startswith("v_cvt_f32_ubyte"):
        return "unpack:v_cvt_f32_ubyteN"
    if op in ("v_sub_f32", "v_subrev_f32"):
        return "unpack:v_sub_f32"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "valu:lane(" + op + ")"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "valu:move"
    if op.startswith(("v_add", "v_sub", "v_lshl", "v_lshr", "v_mad_u", "v_mul_lo", "v_mul_u", "v_and", "v_or", "v_mad_i", "v_ashr", "v_mul_hi", "v_mul_i")):
        return "valu:int/address"
    if op.startswith(("v_cmp", "v_cndmask")):
        return "valu:compare/select"
    if op.startswith(("v_sqrt", "v_exp", "v_rcp", "v_rsq", "v_log")):
        return "valu:transcendental"
    if op.startswith("v_"):
        return "valu:other(" + op + ")"
    if op.startswith("ds_"):
        return "lds:" + op
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem:" + op
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_"):
        return "salu/branch"
    return "other:" + op


def main():
    path, name = sys.argv[1], sys.argv[2]
    min_fma = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
    blocks, cur, label = [], collections.Counter(), "entry"
    first = start
    for i in range(start + 1, end):
        l = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):|^; %bb\.(\d+):", l)
        if m:
            blocks.append((label, first, i, cur))
            cur, label, first = collections.Counter(), m.group(1) or "bb." + m.group(2), i
            continue
        t = l.strip()
        if not t or t.startswith((";", ".")):
            continue
        cur[classify(re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", t.split()[0]))] += 1
    blocks.append((label, first, end, cur))
    for label, a, b, c in blocks:
        nf = c["tap:v_fma_f32"]
        if nf < min_fma:
            continue
        valu = sum(v for k, v in c.items() if k.startswith(("tap:", "unpack:", "valu:")))
        print("block %s (listing lines %d-%d): %d VALU instructions, %d v_fma_f32 + %d v_mul_f32 = %d tap instructions; VALU / tap = %.3f" %
              (label, a - start, b - start, valu, nf, c["tap:v_mul_f32"], nf + c["tap:v_mul_f32"], valu / float(nf + c["tap:v_mul_f32"])))
        for k in sorted(c, key=lambda k: (-c[k], k)):
            print("    %4d  %s" % (c[k], k))


if __name__ == "__main__":
    main()
