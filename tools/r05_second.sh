#!/bin/bash
# round 5, second GPU call: new parity tests (full frame, exact mode), exact-mode cost, config-2 tiles, small-frame timeline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_second; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python tools/time_exact.py > $O/time_exact.txt 2>&1; tail -6 $O/time_exact.txt
timeout 600 python tools/ab_asw.py --only=c2,d64 base "g40j8=SSAMD_ASW_GEOM=40,17,8" "g16j8=SSAMD_ASW_GEOM=16,17,8" "g30j16=SSAMD_ASW_GEOM=30,17,16" "g20j8=SSAMD_ASW_GEOM=20,17,8" "tail1=SSAMD_ASW_TAIL=1" > $O/c2_geom.txt 2>&1; cat $O/c2_geom.txt | head -12
bash tools/trace_small_frame.sh > $O/trace_tsu.txt 2>&1; tail -12 $O/trace_tsu.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.err
echo "== small frames, fused prepass (default)"; timeout 300 python tools/small_frames.py > $O/small_fused.txt 2>&1; cat $O/small_fused.txt
echo "== small frames, SSAMD_ASW_PREPASS_FUSE=0"; SSAMD_ASW_PREPASS_FUSE=0 timeout 300 python tools/small_frames.py > $O/small_unfused.txt 2>&1; cat $O/small_unfused.txt
