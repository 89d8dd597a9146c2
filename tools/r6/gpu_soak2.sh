#!/bin/bash
# round 6: second soak of the final build (fresh seed bases)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "build: $(sha256sum pypore_amd/libporeseg.so | cut -c1-64)  $(date -u +%FT%TZ)" > gpurun_out/r06_soak2.txt
FUZZ_BASE=12000000 timeout 1500 python tools/fuzz_gpu.py 12000 2>&1 | tail -2 | cut -c1-300 >> gpurun_out/r06_soak2.txt
FUZZ_SCALE=64 FUZZ_BASE=12500000 timeout 600 python tools/fuzz_gpu.py 8000 2>&1 | tail -2 | cut -c1-300 >> gpurun_out/r06_soak2.txt
timeout 900 python tools/fuzz_sparse.py 3000 70000 2>&1 | tail -2 | cut -c1-300 >> gpurun_out/r06_soak2.txt
timeout 600 python tools/r6/fuzz_exact.py 3000 70000 2>&1 | tail -2 | cut -c1-300 >> gpurun_out/r06_soak2.txt
timeout 900 python tools/r6/fuzz_single_pass.py 20000 200000 2>&1 | tail -3 | cut -c1-300 >> gpurun_out/r06_soak2.txt
cat gpurun_out/r06_soak2.txt
