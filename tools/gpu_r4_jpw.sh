#!/bin/bash
# round 4: jobs per working slot of the subtree kernel (default 4), five interleaved rounds on the final build
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "single", r["single_stream"]["sequence_ms"], r["kernel_ms"]["tree_ms"])'
for rep in 1 2 3 4 5; do
for v in 4 5 6 8; do
  echo -n "[jpw $v] "; PORESEG_TREE_JPW=$v python bench.py --no-cpu --no-h2d --steps 100 --warmup 20 2>/dev/null | python -c "$P"
done
done
