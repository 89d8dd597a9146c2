#!/bin/bash
# round 4: depth of the row ring (rows requested ahead; _d1 / _d2 / _d3, default 4) and refills only when a live row is left
# (_rgd2): suite of each variant, interleaved A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for lib in ${SUITE_LIBS:-_d1 _d2 _d3}; do
  for env in "X=0" "PORESEG_MODE=2"; do
    echo "== $lib $env"; PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so env $env timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -1
  done
done
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "single", r["single_stream"]["sequence_ms"], {k: v for k, v in r["kernel_ms"].items() if k in ("spine_ms", "tree_ms", "bridge_ms")})'
for rep in 1 2 3 4 5; do
  for lib in ${AB_LIBS:-"" _d1 _d2 _d3 _rgd2}; do
    [ "$lib" = "''" ] && lib=""
    echo -n "[$lib] "; PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so python bench.py --no-cpu --no-h2d --steps 160 --warmup 32 2>/dev/null | python -c "$P"
  done
done
