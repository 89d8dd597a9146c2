#!/bin/bash
# round 4, final build, the bench's default number of calls in flight: runtime switches that were tuned with four (K0 waves
# per SIMD, subtree jobs per wave, share of the wave slots), interleaved rounds; prints ms per step and the lone call
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], r["single_stream"]["sequence_ms"], end=" ")'
for rep in 1 2 3 4 5; do
  for v in ${VARIANTS:-"X=0" "PORESEG_K0_WAVES=4" "PORESEG_TREE_JPW=3" "PORESEG_TREE_JPW=6" "PORESEG_TREE_JPW=8" "PORESEG_SLOTS_PCT=75"}; do
    echo -n "[$v] "; env $v python bench.py --no-cpu --no-h2d --steps 160 --warmup 32 2>/dev/null | python -c "$P"
  done; echo
done
