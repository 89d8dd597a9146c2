#!/bin/bash
# round 4: filter tests (orders 5-8, float64 input), the suite in default and verify mode, the deep-job kernel: off, 4 waves, 8 waves
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_filter.py -x -q -m gpu 2>&1 | tail -15
for env in "X=0" "PORESEG_MODE=2" "PORESEG_TREE_PAR=0"; do
  echo "== $env"; env $env timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
done
echo "== tree_par=0"; PORESEG_TREE_PAR=0 python tools/bench_filtered_parse.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-330
for lib in "" _par8; do
  echo "== lib [$lib]"; PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so python tools/bench_filtered_parse.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-330
done
