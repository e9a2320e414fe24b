#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_soak; mkdir -p $O
cd $R
timeout 400 python tools/soak_r05.py 240 > $O/soak_r05.txt 2>&1; tail -3 $O/soak_r05.txt
timeout 300 python tools/soak_pipe.py 150 > $O/soak_pipe.txt 2>&1; tail -2 $O/soak_pipe.txt
timeout 300 python tools/soak_wave.py 150 > $O/soak_wave.txt 2>&1; tail -2 $O/soak_wave.txt
timeout 300 python tools/soak_threads.py 90 > $O/soak_threads.txt 2>&1; tail -2 $O/soak_threads.txt
