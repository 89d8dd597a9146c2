#!/bin/bash
# round 6: single-pass route after the edge kernel's loads were hoisted: tests, kernel times, bench A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_single_pass.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-250
PORESEG_SCAN_BS=0 timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | cut -c1-300
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
cd /tmp
for sp in 1; do
  rm -rf /tmp/kf
  PORESEG_SINGLE_PASS=$sp rocprofv3 --kernel-trace --stats -d /tmp/kf -o out --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload file --steps 10 --warmup 3 --no-cpu --no-h2d --streams 1 --diag-env > /tmp/kf_$sp.log 2>&1
  tail -1 /tmp/kf_$sp.log | cut -c1-300
  cp $(find /tmp/kf -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/r6_file_sp${sp}_kernel_stats.csv
  head -16 $GRAFT_REPO_ROOT/gpurun_out/r6_file_sp${sp}_kernel_stats.csv | cut -c1-50,150-260
done
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
timeout 600 python bench.py --no-cpu --no-h2d 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
f = d.get('int16_file')
print(d['ms_per_step'], f['ms_per_step'], f['two_calls_ms_per_step'], f['two_calls_same_boundaries'], f['roofline']['frac'])"
done
