#!/bin/bash
# one seed of the scale-64 fuzz (verify-mode disagreement found by the round-4 validation): this build and round 3's
cd "$GRAFT_REPO_ROOT"
echo "== round 4 build"; FUZZ_SCALE=64 FUZZ_BASE=4001427 python tools/fuzz_gpu.py 1 2>&1 | grep -v amdgpu.ids | cut -c1-400
echo "== round 4 build, PORESEG_GROUPS=0 / TREE_PAR=0"; PORESEG_TREE_PAR=0 FUZZ_SCALE=64 FUZZ_BASE=4001427 python tools/fuzz_gpu.py 1 2>&1 | grep -v amdgpu.ids | cut -c1-300
if [ -d _r3wt ]; then cd _r3wt; echo "== round 3 build (9afed96)"; FUZZ_SCALE=64 FUZZ_BASE=4001427 python tools/fuzz_gpu.py 1 2>&1 | grep -v amdgpu.ids | cut -c1-400; fi
