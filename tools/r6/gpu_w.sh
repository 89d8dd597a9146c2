#!/bin/bash
# round 6: admitted K0s M = 2 / 3 / 4 / 6 on the driver's 20 steps and on 100 steps (pool override through --diag-env), interleaved
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], "seq", r["sequence_ms"], d["host"]["poreseg_env_seen"])'
for rep in 1 2 3 4 5; do
  for M in 2 3 4 6; do
    echo -n "[M $M]  20: "; PORESEG_POOL_K0_MAX=$M timeout 300 python bench.py --diag-env --no-cpu --no-h2d --no-detail --steps 20 --warmup 5 2>/dev/null | python -c "$P"
    echo -n "[M $M] 100: "; PORESEG_POOL_K0_MAX=$M timeout 300 python bench.py --diag-env --no-cpu --no-h2d --no-detail --steps 100 --warmup 20 2>/dev/null | python -c "$P"
  done
done 2>&1 | tee gpurun_out/r6_admit_burst_ab.txt
