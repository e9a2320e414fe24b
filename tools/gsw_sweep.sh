#!/bin/bash
# GPU box: StereoGSW config 4 (1080p, D 0..192, win 11) kernel time for forced geometries "XG,DG,Ty[,Hy]"
# (Hy thread groups share the e tile of an image row: strips of Ty * Hy output rows, round 3)
echo "default: $(python tools/time_gsw.py 2>/dev/null)"
for g in "$@"; do
  echo "$g: $(SSAMD_GSW_GEOM=$g python tools/time_gsw.py 2>&1 | tail -1 | cut -c1-200)"
done
