#!/bin/bash
# round 5, final build: K0 waves per SIMD x K0 admission, fp32 trace (100 and 20 steps) and the int16 file workload
cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], end=" ")'
for rep in 1 2; do
for cfg in "2 3" "1 3" "3 3" "4 3" "2 2" "2 4" "2 5" "3 4"; do
  set -- $cfg
  echo -n "[k0_waves $1 admit $2] trace 100: "; PORESEG_K0_WAVES=$1 PORESEG_POOL_K0_MAX=$2 python bench.py --no-cpu --no-h2d --no-detail 2>/dev/null | python -c "$P"
  echo -n " 20: "; for i in 1 2; do PORESEG_K0_WAVES=$1 PORESEG_POOL_K0_MAX=$2 python bench.py --no-cpu --no-h2d --no-detail --steps 20 --warmup 5 2>/dev/null | python -c "$P"; done
  echo -n " file 40: "; PORESEG_K0_WAVES=$1 PORESEG_POOL_K0_MAX=$2 python bench.py --workload file --no-cpu --no-detail --steps 40 --warmup 8 2>/dev/null | python -c "$P"; echo
done
done
