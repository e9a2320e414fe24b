cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for g in "" "40,12,1" "20,25,2" "16,32,2" "32,16,2" "25,20,2" "40,12,2"; do
  echo "== GEOM '$g'"
  if [ -n "$g" ]; then export SSAMD_GSW_GEOM=$g; else unset SSAMD_GSW_GEOM; fi
  python tools/run_asw.py --gsw --win 11 --steps 3 2>&1 | tail -2
done
