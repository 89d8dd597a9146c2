#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
for env in "X=0" "PORESEG_MODE=2"; do
  echo "== $env"; env $env timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
done
python bench.py --no-cpu --no-h2d --steps 100 --warmup 20 2>/dev/null | python -c '
import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "single", r["single_stream"]["sequence_ms"], r["kernel_ms"]); print(d.get("int16_file")); print(r["evaluations_per_s"], d["config"]["ranks_seen"])'
python bench.py --no-cpu --no-h2d --steps 100 --warmup 20 2>/dev/null | python -c '
import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "single", r["single_stream"]["sequence_ms"], r["kernel_ms"])'
python tools/bench_align.py 2>&1 | tail -8
bash tools/pmc_run.sh r4y_pmc 1 2>&1 | grep -E "^sq (void )?ps::(blocksum|spine|tree|bridge)" | cut -c1-260
