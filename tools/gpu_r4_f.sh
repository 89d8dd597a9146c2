#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
for env in "X=0" "PORESEG_MODE=2"; do
  echo "== $env"; env $env timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
done
for t in 40000 30000 20000 16000; do echo "== lone filtered event, tile $t"; TILE=$t BATCH=2 python tools/bench_filtered_parse.py 2>&1 | grep -v amdgpu.ids | sed -n 1,2p | cut -c1-260; done
bash tools/gpu_r4_sweep.sh
