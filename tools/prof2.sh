#!/bin/bash
# GPU box: rocprofv3 evidence for one ASW configuration.  usage: [ENV=.. ] tools/prof2.sh <tag> [run_asw.py args...]
# -> gpurun_out/prof_<tag>/{stats,pmc1..3}/ + summary.txt.  Counter passes are separate runs (PMC never combined with traces).
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$tag; mkdir -p $O
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o s -- python $R/tools/run_asw.py --steps 3 "$@" > $O/stats.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $O/pmc1 -o p -- python $R/tools/run_asw.py --steps 1 "$@" > $O/pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAVES SQ_THREAD_CYCLES_VALU -d $O/pmc2 -o p -- python $R/tools/run_asw.py --steps 1 "$@" > $O/pmc2.log 2>&1
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $O/pmc3 -o p -- python $R/tools/run_asw.py --steps 1 "$@" > $O/pmc3.log 2>&1
python $R/tools/prof_summary.py $O > /dev/null
grep -A12 "asw_aggregate" $O/summary.txt | head -60
