#!/bin/bash
# sparse traces with sixteen calls in flight, helpers on / off; then the lone-call regimes again
out=gpurun_out/r05_lat_help3.txt
{
for dw in "2e8 3e8" "1e7 5e7" "1e5 1e6"; do
  set -- $dw
  for on in pool 0; do
    if [ $on = pool ]; then unset PORESEG_LAT_HELP; else export PORESEG_LAT_HELP=$on; fi; DWELL_LO=$1 DWELL_HI=$2 timeout 300 python tools/bound_probe.py 16 64 2>&1 | tail -1 | sed "s/^/dwell $1-$2 helpers $on: /" | cut -c1-160
  done
done
unset PORESEG_LAT_HELP; timeout 600 python tools/dbg_lat_help.py 2>&1 | grep -v amdgpu.ids | tail -22
} > $out 2>&1
cat $out
