#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_tu; mkdir -p $O
cd $R; E=$R/tools/_exp
CASES="--cases=tsu:288:384:16:0:15:0,tsu35c:288:384:16:0:35:1,vga16:480:640:16:0:35:0,d12:1080:1920:12:0:35:0,d16w11:1080:1920:16:0:11:0,d19:1080:1920:19:0:35:0"
timeout 600 python tools/ab_asw.py $CASES "u3=SSAMD_LIB=$E/libssamd_units3.so" "u4=SSAMD_LIB=$R/simplestereo_amd/libssamd.so" "u3b=SSAMD_LIB=$E/libssamd_units3.so" "u4b=SSAMD_LIB=$R/simplestereo_amd/libssamd.so" > $O/tu_ab3.txt 2>&1; head -9 $O/tu_ab3.txt
