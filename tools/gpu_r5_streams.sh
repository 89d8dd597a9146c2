#!/bin/bash
# round 5: contexts in flight x hardware queues x K0 admission, the driver's 20 steps (three runs) and the default 100
cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], end=" ")'
for rep in 1 2; do
for cfg in "16 16 3" "16 24 3" "20 24 3" "24 24 3" "24 32 3" "32 32 3" "20 24 4" "24 32 4" "16 24 0"; do
  set -- $cfg
  echo -n "[streams $1 queues $2 admit $3] 100: "; GPU_MAX_HW_QUEUES=$2 PORESEG_POOL_K0_MAX=$3 python bench.py --no-cpu --no-h2d --no-detail --streams $1 2>/dev/null | python -c "$P"
  echo -n " 20: "; for i in 1 2 3; do GPU_MAX_HW_QUEUES=$2 PORESEG_POOL_K0_MAX=$3 python bench.py --no-cpu --no-h2d --no-detail --streams $1 --steps 20 --warmup 5 2>/dev/null | python -c "$P"; done; echo
done
done
