#!/bin/bash
# round 6: kernel times of the file route (config 3, one call at a time) in one pass and by two calls
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
cd /tmp
for sp in 1 0; do
  rm -rf /tmp/kf
  PORESEG_SINGLE_PASS=$sp rocprofv3 --kernel-trace --stats -d /tmp/kf -o out --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload file --steps 20 --warmup 3 --no-cpu --no-h2d --streams 1 --diag-env > /tmp/kf_$sp.log 2>&1
  tail -1 /tmp/kf_$sp.log | cut -c1-300
  cp $(find /tmp/kf -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/r06_file_sp${sp}_kernel_stats.csv
  head -8 $GRAFT_REPO_ROOT/gpurun_out/r06_file_sp${sp}_kernel_stats.csv | cut -c1-50,150-260
done
