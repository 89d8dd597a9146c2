#!/bin/bash
# round 6: the pool's step driven by native threads (no Python in the loop) against python bench.py on the same box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16 LD_LIBRARY_PATH=$PWD/pypore_amd:$LD_LIBRARY_PATH
[ -x tools/probes/pool_native ] || /opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/probes/pool_native.cpp -Iinclude -Lpypore_amd -lporeseg -Wl,-rpath,$PWD/pypore_amd -lpthread -o tools/probes/pool_native 2>/dev/null
for i in 1 2; do
  timeout 300 tools/probes/pool_native 16 100 32
  timeout 300 tools/probes/pool_native 16 20 5 | tail -2
  timeout 600 python bench.py --no-cpu --no-h2d --no-detail 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('python bench 100 steps', d['ms_per_step'])"
  timeout 600 python bench.py --no-cpu --no-h2d --no-detail --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('python bench 20 steps', d['ms_per_step'])"
done
