#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "build: $(sha256sum pypore_amd/libporeseg.so | cut -c1-16)  $(date -u +%FT%TZ)"
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
timeout 900 python tools/r6/option_ab_probe.py base gather_fused=0 download_by_kernel=0 gather_fused=0,download_by_kernel=0 --pairs 8 2>&1 | tail -5 | tee gpurun_out/r6_gather_ab.txt
