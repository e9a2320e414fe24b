#!/bin/bash
# round 6: the N > 1 bench path launched the way the driver launches a scaling run, every rank on GPU 0 over gloo (SSAMD_BENCH_SHARE_GPU=1: a test
# hook the line flags -- NOT a scaling measurement); keeps the compact stdout line and the full record of each N
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_share; mkdir -p $O
cd $R
for n in 8 4 2; do
SSAMD_BENCH_DETAILS=$O/details_$n.json SSAMD_BENCH_SHARE_GPU=1 OMP_NUM_THREADS=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2971$n bench.py --gpus $n --steps 4 --warmup 1 > $O/line_$n.json 2> $O/err_$n.log; echo "share $n rc=$? bytes=$(wc -c < $O/line_$n.json)"
done
cat $O/line_8.json
