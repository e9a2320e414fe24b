cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_tsu; mkdir -p $O
SSAMD_AUTOTUNE=0 rocprofv3 --output-format csv --kernel-trace -d $O -o t -- python $R/tools/run_asw.py --h 288 --w 384 --maxd 16 --win 15 --steps 6 > $O/log.txt 2>&1
python - "$O" <<'PY'
import csv, glob, sys, os
O=sys.argv[1]
f=glob.glob(os.path.join(O,'**','*kernel_trace.csv'), recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
prev_end=None
for r in rows[-12:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print('%-60s start %9.1f us dur %7.1f us gap %6.1f us' % (r['Kernel_Name'][:60], (s-t0)/1e3, (e-s)/1e3, (s-prev_end)/1e3 if prev_end else 0))
    prev_end=e
PY
