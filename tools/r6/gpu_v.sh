#!/bin/bash
# round 6: contexts in flight 16 / 20 / 24 (hardware queues to match) on the 100-step and the 20-step command, interleaved
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], "seq", r["sequence_ms"])'
for rep in 1 2 3 4; do
  for T in 16 20 24; do
    Q=$T
    echo -n "[T $T Q $Q] 100: "; GPU_MAX_HW_QUEUES=$Q timeout 300 python bench.py --no-cpu --no-h2d --no-detail --steps 100 --warmup 20 --streams $T 2>/dev/null | python -c "$P"
    echo -n "[T $T Q $Q]  20: "; GPU_MAX_HW_QUEUES=$Q timeout 300 python bench.py --no-cpu --no-h2d --no-detail --steps 20 --warmup 5 --streams $T 2>/dev/null | python -c "$P"
  done
done 2>&1 | tee gpurun_out/r6_streams_ab.txt
