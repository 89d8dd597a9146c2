#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "build: $(sha256sum pypore_amd/libporeseg.so | cut -c1-16)  $(date -u +%FT%TZ)"
timeout 1200 python -u tools/r6/option_ab_probe.py base k0_sets=3,k0_waves=2,k0_admit=1 k0_sets=4,k0_waves=2,k0_admit=1 k0_sets=2,k0_waves=3,k0_admit=1 k0_sets=3,k0_waves=1,k0_admit=2 k0_sets=2,k0_waves=2,k0_admit=2 --pairs 6 2>&1 | tail -7 | tee gpurun_out/r6_k0_sets_ab2.txt
