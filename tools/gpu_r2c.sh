#!/bin/bash
# round 2, run C: every bench workload once (JSON lines into gpurun_out/r2c_*.json)
mkdir -p gpurun_out
for w in trace file sharded-trace files; do
  echo "== $w"
  ( time timeout 900 python bench.py --workload $w > gpurun_out/r2c_$w.json 2> gpurun_out/r2c_$w.err ) 2>&1 | grep real
  tail -c 3000 gpurun_out/r2c_$w.json; echo; tail -3 gpurun_out/r2c_$w.err | grep -v amdgpu.ids
done
nproc; free -g | head -2
