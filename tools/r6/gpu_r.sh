#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16 PORESEG_LIB=$GRAFT_REPO_ROOT/pypore_amd/libporeseg_diag.so
cd /tmp
for rs in 1; do
  rm -rf /tmp/kf
  PORESEG_REP_STAGE=$rs rocprofv3 --kernel-trace --stats -d /tmp/kf -o out --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload file --steps 10 --warmup 3 --no-cpu --no-h2d --streams 1 --diag-env > /tmp/kf_$rs.log 2>&1
  tail -1 /tmp/kf_$rs.log | cut -c1-200
  grep "edge_cls\|blocksum" $(find /tmp/kf -name '*kernel_stats.csv' | head -1) | cut -c1-40,150-300
done
cd "$GRAFT_REPO_ROOT"; unset PORESEG_LIB
timeout 600 python -m pytest tests/test_single_pass.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-250
for i in 1 2 3; do
timeout 600 python bench.py --no-cpu --no-h2d 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
f = d.get('int16_file')
print(d['ms_per_step'], f['ms_per_step'], f['two_calls_ms_per_step'], f['two_calls_same_boundaries'], f['roofline']['frac'])"
done
