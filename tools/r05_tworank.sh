#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_tworank; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_exact.py tests/test_gpu_strips_multiprocess.py tests/test_gpu_rccl_world1.py -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for n in 2 4; do
SSAMD_BENCH_SHARE_GPU=1 OMP_NUM_THREADS=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 6 --warmup 2 --no-bad1 > $O/bench_share_$n.json 2> $O/bench_share_$n.err; echo "share $n rc=$?"
SSAMD_STRIP_OVERLAP=0 SSAMD_BENCH_SHARE_GPU=1 OMP_NUM_THREADS=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2952$n bench.py --gpus $n --steps 6 --warmup 2 --no-bad1 > $O/bench_share_${n}_seq.json 2> $O/bench_share_${n}_seq.err; echo "share $n sequential rc=$?"
done
