#!/bin/bash
# usage: tools/prof.sh <tag> [run_asw.py args...]   (GPU box; writes gpurun_out/prof_<tag>/)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$tag; mkdir -p $O
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o s -- python $R/tools/run_asw.py "$@" > $O/stats.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $O/pmc1 -o p -- python $R/tools/run_asw.py --steps 1 "$@" > $O/pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAVES SQ_INSTS_FLAT -d $O/pmc2 -o p -- python $R/tools/run_asw.py --steps 1 "$@" > $O/pmc2.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc3 -o p -- python $R/tools/run_asw.py --steps 1 "$@" > $O/pmc3.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmc4 -o p -- python $R/tools/run_asw.py --steps 1 "$@" > $O/pmc4.log 2>&1
find $O -name "*.csv" | head -30
python $R/tools/prof_summary.py $O
