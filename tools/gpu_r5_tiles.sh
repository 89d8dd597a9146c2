#!/bin/bash
# round 5, final build: tile length (speculative windows against chains and occupancy) and subtree jobs per working wave,
# sixteen calls in flight; windows per call beside the step time
cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["work"]["windows"], d["work"]["tiles"], end=" | ")'
for rep in 1 2 3; do
for v in "X=0" "PORESEG_TILE=81920" "PORESEG_TILE=98304" "PORESEG_TILE=131072" "PORESEG_TILE=196608" "PORESEG_TREE_JPW=6" "PORESEG_TREE_JPW=8"; do
  echo -n "[$v] "; env $v python bench.py --no-cpu --no-h2d --no-detail 2>/dev/null | python -c "$P"; echo
done
done
