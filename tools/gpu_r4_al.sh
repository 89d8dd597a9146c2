#!/bin/bash
# round 4: rows of the sweep aligned to pairs of groups (library variant _al) against the default geometry
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
export PORESEG_LIB=$PWD/pypore_amd/libporeseg_al.so
timeout 900 python -m pytest tests/test_bound_audit.py -x -q -m gpu 2>&1 | tail -5
for env in "X=0" "PORESEG_MODE=2"; do
  echo "== _al $env"; env $env timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
done
unset PORESEG_LIB
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "single", r["single_stream"]["sequence_ms"] if r["single_stream"] else None, {k: v for k, v in r["kernel_ms"].items() if k in ("spine_ms", "bridge_ms", "tree_ms")})'
for rep in 1 2 3; do
  for lib in "" _al; do
    echo -n "[$lib] "; PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so python bench.py --no-cpu --no-h2d --steps 100 --warmup 20 2>/dev/null | python -c "$P"
  done
done
PORESEG_LIB=$PWD/pypore_amd/libporeseg_al.so bash tools/pmc_run.sh r4al_pmc 1 2>&1 | grep -E "^sq (void )?ps::(blocksum|spine|tree|bridge)" | cut -c1-260
