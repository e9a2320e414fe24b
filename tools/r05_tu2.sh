#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_tu; mkdir -p $O
cd $R; E=$R/tools/_exp
timeout 900 python tools/ab_asw.py --only=c3,c5,c2,d64 "m0=SSAMD_LIB=$E/libssamd_m0.so" "m_bias0=SSAMD_LIB=$E/libssamd_m_bias0.so" "m_bias50=SSAMD_LIB=$E/libssamd_m_bias50.so" "m_trk=SSAMD_LIB=$E/libssamd_m_trk.so" "m_nounc=SSAMD_LIB=$E/libssamd_m_nounc.so" "m_nocl=SSAMD_LIB=$E/libssamd_m_nocl.so" "m_O2=SSAMD_LIB=$E/libssamd_m_O2.so" "m_nopost=SSAMD_LIB=$E/libssamd_m_nopost.so" "m0b=SSAMD_LIB=$E/libssamd_m0.so" > $O/tu_ab2.txt 2>&1; head -6 $O/tu_ab2.txt
