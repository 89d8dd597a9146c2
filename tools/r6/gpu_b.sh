#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "build: $(sha256sum pypore_amd/libporeseg.so | cut -c1-16) diag $(sha256sum pypore_amd/libporeseg_diag.so | cut -c1-16)  $(date -u +%FT%TZ)"
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -8
PORESEG_LIB=$PWD/pypore_amd/libporeseg_diag.so timeout 900 python tools/r6/k0_grp_probe.py 16 100 12 2>&1 | tail -8 | tee gpurun_out/r6_k0_grp_probe.txt
