#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "build: $(sha256sum pypore_amd/libporeseg.so | cut -c1-16)  $(date -u +%FT%TZ)"
timeout 1200 python -u tools/r6/option_ab_probe.py base k0_shared=1,k0_waves=0,k0_admit=0 k0_shared=1 cu_split=64 --pairs 3 2>&1 | tail -6 | tee -a gpurun_out/r6_cu_split_ab.txt
