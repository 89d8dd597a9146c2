#!/bin/bash
# round 4: per-kernel times of one call at a time, then the in-kernel phase stamps (diagnostic build) of the same calls
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "seq", r["sequence_ms"], r["kernel_ms"], d["work"])'
for g in ${GROUPS_LIST:-1 0}; do
  echo "== groups=$g, one call at a time"
  PORESEG_GROUPS=$g python bench.py --no-cpu --no-h2d --steps 40 --warmup 10 --streams 1 2>/dev/null | python -c "$P"
  PORESEG_GROUPS=$g PORESEG_LIB=$PWD/pypore_amd/libporeseg_stamp.so python bench.py --no-cpu --no-h2d --no-detail --steps 2 --warmup 1 --streams 1 2>&1 | grep "poreseg stamps" | tail -7
done
