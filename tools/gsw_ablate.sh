for g in 20,25,2,1 10,25,2,2; do
  for v in "" gnoe gnow gnotaps gnoew; do
    if [ -z "$v" ]; then echo "$g full: $(SSAMD_GSW_GEOM=$g python tools/time_gsw.py 2>&1 | tail -1 | cut -c60-140)";
    else echo "$g $v: $(SSAMD_GSW_GEOM=$g SSAMD_EXPERIMENT=1 SSAMD_LIB=tools/_exp/libssamd_$v.so python tools/time_gsw.py 2>&1 | tail -1 | cut -c60-140)"; fi
  done
done
