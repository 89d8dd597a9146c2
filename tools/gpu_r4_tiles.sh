#!/bin/bash
# round 4: tile length of the 1e8-sample trace, five interleaved rounds (default: total / 2048 = 48 832 samples)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "single", r["single_stream"]["sequence_ms"], d["work"]["windows"])'
for rep in 1 2 3 4 5; do
for v in "X=0" "PORESEG_TILE=57344" "PORESEG_TILE=65536" "PORESEG_TILE=81920"; do
  echo -n "[$v] "; env $v python bench.py --no-cpu --no-h2d --no-detail --steps 100 --warmup 20 2>/dev/null | python -c "$P"
done
done
echo "== filtered event with the queue fix"; python tools/bench_filtered_parse.py 2>&1 | grep -v amdgpu.ids | sed -n 2,3p | cut -c1-200
