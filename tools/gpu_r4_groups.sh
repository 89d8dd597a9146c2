#!/bin/bash
# round 4: the coarse pass over the group records (PORESEG_GROUPS=1, default) against the sweep of every row (=0):
# GPU suite in default and verify mode, then interleaved bench rounds of both settings on one box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for env in "X=0" "PORESEG_MODE=2" "PORESEG_GROUPS=0"; do
  echo "== $env"; env $env timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
done
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "seq", r["sequence_ms"], "single", r["single_stream"]["sequence_ms"], r["single_stream"].get("kernel_ms"))'
for rep in 1 2 3; do
  for g in 0 1; do
    echo -n "[groups=$g] "; PORESEG_GROUPS=$g python bench.py --no-cpu --no-h2d --steps 100 --warmup 20 2>/dev/null | python -c "$P"
  done
done
