#!/bin/bash
# rocprofv3 kernel stats of the trace without steps, helpers of the look-ahead kernel off / on
export TMPDIR=/tmp; cd /tmp
for on in 0 1; do
  rm -rf /tmp/kst$on
  rocprofv3 --kernel-trace --stats -d /tmp/kst$on -o out --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dbg_flat_timing.py $on > /tmp/kst$on.log 2>&1
  cp $(find /tmp/kst$on -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/r05_flat_kernel_stats_helpers$on.csv
  tail -2 /tmp/kst$on.log | cut -c1-300
done
