#!/bin/bash
# validation of a kernel change: GPU suite in default / verify / single-wave-tree / LDS-window / host-stitch modes,
# then the randomised fuzz (default + verify) and the regime parity sweep.  usage: gpu_validate.sh [fuzz seeds]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
N=${1:-3000}
# (round 4: the single-wave subtree kernel is the default, so the third mode is the multi-wave one; a sixth and seventh run
#  switch the coarse pass and the shared deep jobs off)
echo "build: $(sha256sum pypore_amd/libporeseg.so | cut -c1-16)  $(date -u +%FT%TZ)"
for env in "X=0" "PORESEG_MODE=2" "PORESEG_TREE_MW=1" "PORESEG_SCAN_BS=0" "PORESEG_STITCH=host" "PORESEG_GROUPS=0" "PORESEG_TREE_PAR=0"; do
  echo "== $env"; env $env timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
done
FUZZ_BASE=${FUZZ_BASE:-2000000} timeout 3000 python tools/fuzz_gpu.py $N 2>&1 | tail -2 | cut -c1-420
FUZZ_SCALE=64 FUZZ_BASE=${FUZZ_BASE:-2000000} timeout 3000 python tools/fuzz_gpu.py $((N / 2)) 2>&1 | tail -3 | cut -c1-520
timeout 900 python tools/regime_parity.py 2>&1 | tail -3
timeout 600 python tools/fuzz_many_events.py 2>&1 | tail -2
