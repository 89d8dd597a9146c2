#!/bin/bash
# round 6: the validation once more at the end of the round (library unchanged: same SHA-256; the suite has grown by three tests)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash tools/gpu_validate.sh 3000 > gpurun_out/r06_validation.txt 2>&1
grep -E "^build|^==|passed|failed|problems|mismatching" gpurun_out/r06_validation.txt | cut -c1-160
