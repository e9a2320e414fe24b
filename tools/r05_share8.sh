#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_share8; mkdir -p $O
cd $R
for n in 8 4; do
SSAMD_BENCH_SHARE_GPU=1 OMP_NUM_THREADS=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2961$n bench.py --gpus $n --steps 4 --warmup 1 > $O/bench_share_$n.json 2> $O/bench_share_$n.err; echo "share $n rc=$?"; tail -c 300 $O/bench_share_$n.err
done
