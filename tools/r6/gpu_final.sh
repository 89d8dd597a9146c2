#!/bin/bash
# round 6, final build: the round's profile collection, the other workloads' bench lines, the validation and a soak
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
python bench.py --workload sharded-trace --no-cpu > gpurun_out/r06_sharded.json 2>> gpurun_out/r06_bench.err
python bench.py --workload files --files 16 > gpurun_out/r06_files16.json 2>> gpurun_out/r06_bench.err
python bench.py --workload file --no-cpu > gpurun_out/r06_file.json 2>> gpurun_out/r06_bench.err
bash tools/r6/gpu_file_profile.sh > gpurun_out/r06_file_profile.log 2>&1
bash tools/gpu_validate.sh 3000 > gpurun_out/r06_validation.txt 2>&1
tail -14 gpurun_out/r06_validation.txt | cut -c1-300
FUZZ_BASE=9600000 bash tools/gpu_soak.sh 6000 4000 > gpurun_out/r06_soak_final.txt 2>&1
timeout 900 python tools/r6/fuzz_single_pass.py 3000 50000 >> gpurun_out/r06_soak_final.txt 2>&1
tail -9 gpurun_out/r06_soak_final.txt | cut -c1-300
