#!/bin/bash
# round 4: BASELINE config 5 (one 1e9-sample trace on one GPU) and config 2 on the final build; tile lengths for the long trace
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(d["config"]["name"], d["ms_per_step"], d["value"], r["frac"], r["kernel_ms"], d["config"]["boundaries"], d["config"]["checks"], d["work"]["tiles"])'
for v in "X=0" "PORESEG_TILE=325520" "PORESEG_TILE=488288" "X=0"; do
  echo -n "[$v] "; env $v python bench.py --no-cpu --workload sharded-trace --steps 10 --warmup 3 2>/dev/null | python -c "$P"
done
python tools/bench_config2.py 2>&1 | grep -v amdgpu | tail -3 | cut -c1-300
python tools/bench_experiment.py 2>&1 | grep -v amdgpu | tail -6 | cut -c1-300
