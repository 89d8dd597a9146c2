#!/bin/bash
# round 5: K0 admission gate (PORESEG_K0_MAX calls with their K0 in flight), default 100 steps and the driver's 20
cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], end=" ")'
for rep in 1 2 3; do
for v in "X=0" "PORESEG_K0_MAX=1" "PORESEG_K0_MAX=2" "PORESEG_K0_MAX=3" "PORESEG_K0_MAX=4" "PORESEG_K0_MAX=6" "PORESEG_K0_MAX=8"; do
  echo -n "[$v] 100: "; env GPU_MAX_HW_QUEUES=24 $v python bench.py --no-cpu --no-h2d --no-detail 2>/dev/null | python -c "$P"
  echo -n " 20: "; for i in 1 2 3; do env GPU_MAX_HW_QUEUES=24 $v python bench.py --no-cpu --no-h2d --no-detail --steps 20 --warmup 5 2>/dev/null | python -c "$P"; done; echo
done
done
