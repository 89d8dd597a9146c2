#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
PORESEG_LIB=$PWD/pypore_amd/libporeseg_diag_prio.so timeout 1200 python -u tools/r6/option_ab_probe.py base k0_sets=4,k0_admit=1 k0_sets=4,k0_admit=2 k0_sets=4,k0_waves=2,k0_admit=1 k0_sets=3,k0_admit=2 k0_admit=2 --pairs 4 2>&1 | tail -7 | tee gpurun_out/r6_k0_sets_prio_ab.txt
