#!/bin/bash
# round 4: the three parts of tools/gpu_validate.sh that the first run left open (its record: test assertions that only fit
# the default pipeline stopped the LDS-window and host-stitch suites; the new cross-route check of the scale-64 fuzz
# counted calls that scan nothing as mismatches).  Same library.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "build: $(sha256sum pypore_amd/libporeseg.so | cut -c1-16)  $(date -u +%FT%TZ)"
for env in "PORESEG_SCAN_BS=0" "PORESEG_STITCH=host"; do
  echo "== $env"; env $env timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
done
FUZZ_SCALE=64 FUZZ_BASE=4000000 timeout 3000 python tools/fuzz_gpu.py 5000 2>&1 | tail -3 | cut -c1-640
