#!/bin/bash
# round 4: after the coarse pass the windows are lighter -- are the tile length and the subtree slot rule still right?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "single", r["single_stream"]["sequence_ms"] if r["single_stream"] else None, {k: v for k, v in r["kernel_ms"].items() if k in ("spine_ms", "bridge_ms", "tree_ms")}, d["work"]["windows"])'
for rep in 1 2; do
for v in "X=0" "PORESEG_TREE_JPW=2" "PORESEG_TREE_JPW=3" "PORESEG_TREE_JPW=6" "PORESEG_TREE_JPW=0" "PORESEG_TILE=32768" "PORESEG_TILE=24576" "PORESEG_TILE=65536"; do
  echo -n "[$v] "; env $v python bench.py --no-cpu --no-h2d --steps 100 --warmup 20 2>/dev/null | python -c "$P"
done
done
