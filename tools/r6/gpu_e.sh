#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "build: diag $(sha256sum pypore_amd/libporeseg_diag.so | cut -c1-16)  $(date -u +%FT%TZ)"
PORESEG_LIB=$PWD/pypore_amd/libporeseg_diag.so timeout 240 python -u tools/r6/residency_probe.py 16 60 2>&1 | tail -32 | tee gpurun_out/r6_residency_probe.txt
