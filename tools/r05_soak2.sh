#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_soak; mkdir -p $O
cd $R
timeout 500 python tools/soak_r05.py 300 > $O/soak_r05.txt 2>&1; grep -v amdgpu $O/soak_r05.txt | cut -c1-400 | tail -40
timeout 300 python tools/ab_asw.py --only=c3,c2,d16,tsu,c5 base "ilp=SSAMD_LIB=$R/tools/_exp/libssamd_ilp.so" "memcl=SSAMD_LIB=$R/tools/_exp/libssamd_memcl.so" "bias100=SSAMD_LIB=$R/tools/_exp/libssamd_bias100.so" "nopost=SSAMD_LIB=$R/tools/_exp/libssamd_nopost.so" > $O/flags_ab.txt 2>&1; cat $O/flags_ab.txt | head -12
