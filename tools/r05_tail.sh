#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05_tail
timeout 600 python tools/ab_asw.py --only=c3,c3c,d128w21 base "tail1=SSAMD_ASW_TAIL=1" base2 "tail1b=SSAMD_ASW_TAIL=1" 2>&1 | grep -v amdgpu | tee gpurun_out/r05_tail/tail_ab.txt | head -5
