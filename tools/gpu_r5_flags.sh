#!/bin/bash
# round 5: compiler-flag variants of the library (-O2, -Os, -fno-unroll-loops) against the default -O3: step time, lone call
cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], r["single_stream"]["sequence_ms"], d["config"]["checks"].get("g7_sha256_equal"), end=" | ")'
for rep in 1 2 3; do
for lib in "" ${LIBS:-_o2 _os _nu}; do
  echo -n "[lib$lib] "; PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so python bench.py --no-cpu --no-h2d --no-detail 2>/dev/null | python -c "$P"; echo
done
done
