#!/bin/bash
# GPU box: time a workload for several forced tile shapes (XG,DG).  usage: sweep_geom.sh "<run_asw args>" geom...
ARGS=$1; shift
for g in "$@"; do
  if [ "$g" = "auto" ]; then unset SSAMD_ASW_GEOM; else export SSAMD_ASW_GEOM=$g; fi
  echo "== $ARGS geom $g: $(timeout 120 python tools/run_asw.py --steps 4 $ARGS 2>&1 | grep -E '^step|rror' | sort -k3 -n | head -1)"
done
