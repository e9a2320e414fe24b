#!/bin/bash
# round 5: A/B of LLVM AMDGPU scheduling strategies (whole-library builds, tools/build_variants.sh) on the headline configurations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_flags; mkdir -p $O
cd $R
E=$R/tools/_exp
timeout 900 python tools/ab_asw.py --only=c3,c5,c2,d16,d7,tsu,d64,d32 base "memcl=SSAMD_LIB=$E/libssamd_memcl.so" "ilp=SSAMD_LIB=$E/libssamd_ilp.so" "iterilp=SSAMD_LIB=$E/libssamd_iterilp.so" "itermaxocc=SSAMD_LIB=$E/libssamd_itermaxocc.so" "iterminreg=SSAMD_LIB=$E/libssamd_iterminreg.so" "memcl2=SSAMD_LIB=$E/libssamd_memcl.so" "base2=SSAMD_LIB=$R/simplestereo_amd/libssamd.so" > $O/flags_ab2.txt 2>&1; head -12 $O/flags_ab2.txt
python tools/time_gsw.py > $O/gsw_base.txt 2>&1; tail -3 $O/gsw_base.txt
for v in memcl ilp; do SSAMD_EXPERIMENT=1 SSAMD_LIB=$E/libssamd_$v.so python tools/time_gsw.py > $O/gsw_$v.txt 2>&1; echo $v; tail -3 $O/gsw_$v.txt; done
