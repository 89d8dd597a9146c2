#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
PORESEG_LIB=$PWD/pypore_amd/libporeseg_diag.so timeout 300 python -u tools/r6/cu_mask_probe.py 2>&1 | tail -14 | tee gpurun_out/r6_cu_mask_probe.txt
