#!/bin/bash
# round 4: quick parity (default + verify) of the current build, interleaved A/B of library variants, optional PMC passes
# usage: gpu_r4_ab.sh "<variant suffixes, '' = default>" [pmc]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
for env in "X=0" "PORESEG_MODE=2"; do
  echo "== $env"; env $env timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
done
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "seq", r["sequence_ms"], "single", r["single_stream"]["sequence_ms"], r["kernel_ms"])'
for rep in 1 2 3; do
  for lib in $1; do
    [ "$lib" = "''" ] && lib=""
    echo -n "[$lib] "; PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so python bench.py --no-cpu --no-h2d --steps 100 --warmup 20 2>/dev/null | python -c "$P"
  done
done
if [ "$2" = "pmc" ]; then bash tools/pmc_run.sh r4x_pmc 1 2>&1 | tail -40; fi
