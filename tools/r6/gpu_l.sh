#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -u tools/r6/lone_tile_probe.py 2>&1 | tail -15 | tee gpurun_out/r6_lone_tile_probe.txt
