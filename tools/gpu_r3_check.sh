#!/bin/bash
# Round 3 kernel check: smoke, the GPU parity suite (default and verify mode), then the A/B bench lines.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python __graft_entry__.py smoke > gpurun_out/r3_smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r3_smoke.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r3_pytest.txt
if [ "$1" = "verify" ]; then
  PORESEG_MODE=2 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3_pytest_verify.txt 2>&1; echo "verify rc=$?"; tail -5 gpurun_out/r3_pytest_verify.txt
fi
bash tools/gpu_ab.sh "" > gpurun_out/r3_ab.txt 2>&1; cat gpurun_out/r3_ab.txt
