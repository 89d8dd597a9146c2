#!/bin/bash
# event length against step time, sixteen batches of 5.12e7 samples in flight: whole call / scans alone / K0 alone
out=gpurun_out/r05_c2_probe2.txt
{
for shape in "4096 12500" "1024 50000" "256 200000" "64 800000" "8 6400000" "1 51200000"; do
  for ph in 0 1 2; do PORESEG_DBG_PHASE=$ph python tools/bound_probe_config2.py 16 160 $shape 2>&1 | tail -1 | cut -c1-420; done
done
} > $GRAFT_REPO_ROOT/$out 2>&1
cat $GRAFT_REPO_ROOT/$out
