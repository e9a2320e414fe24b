#!/bin/bash
# round 5: the library with the phase-shifted and the six-per-lane wave kernel in translation units of their own (max-memory-clause /
# max-ilp scheduling) against the single-translation-unit build of the same sources; then the whole gpu tier on it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_tu; mkdir -p $O
cd $R
S=$R/tools/_exp/libssamd_single.so
timeout 900 python tools/ab_asw.py --only=c3,c3c,c5,c2,d16,d16c,d7,tsu,d64,d128w21 "single=SSAMD_LIB=$S" "units=SSAMD_LIB=$R/simplestereo_amd/libssamd.so" "single2=SSAMD_LIB=$S" "units2=SSAMD_LIB=$R/simplestereo_amd/libssamd.so" > $O/tu_ab.txt 2>&1; head -13 $O/tu_ab.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
