#!/bin/bash
# round 4: second form of K0's group amplitudes (variant _amp2): audit, suite, interleaved A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
export PORESEG_LIB=$PWD/pypore_amd/libporeseg_amp2.so
timeout 900 python -m pytest tests/test_bound_audit.py -x -q -m gpu 2>&1 | tail -4
for env in "X=0" "PORESEG_MODE=2"; do
  echo "== _amp2 $env"; env $env timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
done
unset PORESEG_LIB
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "single", r["single_stream"]["sequence_ms"], {k: v for k, v in r["kernel_ms"].items() if k in ("blocksum_ms", "spine_ms", "tree_ms")})'
for rep in 1 2 3 4 5; do
  for lib in "" _amp2; do
    echo -n "[$lib] "; PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so python bench.py --no-cpu --no-h2d --steps 100 --warmup 20 2>/dev/null | python -c "$P"
  done
done
PORESEG_LIB=$PWD/pypore_amd/libporeseg_amp2.so bash tools/pmc_run.sh r4amp_pmc 1 2>&1 | grep -E "^sq (void )?ps::(blocksum|spine|tree)" | cut -c1-260
