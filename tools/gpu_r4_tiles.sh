#!/bin/bash
# round 4, final build: tile length of the 1e8-sample trace (default rule: total / 1536 = 65 104 samples), interleaved rounds
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "single", r["single_stream"]["sequence_ms"], d["work"]["windows"], end=" | ")'
for rep in 1 2 3 4 5; do
for v in "X=0" "PORESEG_TILE=49152" "PORESEG_TILE=40960" "PORESEG_TILE=32768" "PORESEG_TILE=98304"; do
  echo -n "[$v] "; env $v python bench.py --no-cpu --no-h2d --no-detail --steps 100 --warmup 20 2>/dev/null | python -c "$P"
done; echo
done
