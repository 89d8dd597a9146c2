#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "build: $(sha256sum pypore_amd/libporeseg.so | cut -c1-16)  $(date -u +%FT%TZ)"
PORESEG_K0_WAVES=1 PORESEG_K0_SETS=4 timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
timeout 900 python -u tools/r6/option_ab_probe.py base k0_sets=4 k0_sets=4,k0_admit=1 k0_sets=4,k0_admit=2 k0_sets=4,k0_admit=0 --pairs 5 2>&1 | tail -6 | tee gpurun_out/r6_k0_sets_ab.txt
