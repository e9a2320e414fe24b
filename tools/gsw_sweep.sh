#!/bin/bash
echo "default: $(python tools/time_gsw.py 2>/dev/null)"
for g in "20,25,2" "16,25,2" "24,20,2" "20,17,2" "16,17,2" "32,13,2" "20,13,2" "10,25,2" "12,25,2" "28,17,2" "40,10,2" "20,25,1" "16,13,2" "25,20,2" "18,25,2" "22,22,2"; do
  echo "$g: $(SSAMD_GSW_GEOM=$g python tools/time_gsw.py 2>&1 | tail -1 | cut -c1-160)"
done
