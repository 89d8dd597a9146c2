#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "build: $(sha256sum pypore_amd/libporeseg.so | cut -c1-16)  $(date -u +%FT%TZ)"
timeout 1200 python -u tools/r6/option_ab_probe.py base k0_chain=0 k0_admit=2 k0_admit=1 k0_admit=4 k0_admit=6 --pairs 5 2>&1 | tail -7 | tee gpurun_out/r6_k0_chain_ab.txt
PORESEG_LIB=$PWD/pypore_amd/libporeseg_diag.so timeout 1200 python -u tools/r6/option_ab_probe.py base k0_sets=4,k0_admit=1 k0_sets=4,k0_admit=2 k0_sets=3,k0_admit=2 k0_sets=4,k0_waves=2,k0_admit=1 --pairs 4 2>&1 | tail -6 | tee -a gpurun_out/r6_k0_chain_ab.txt
