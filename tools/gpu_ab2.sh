#!/bin/bash
# A/B of library variants x environment settings on the bench trace, two rounds:
#   gpu_ab2.sh "<suffix list>" "<env list>"     e.g.  gpu_ab2.sh "'' _d4" "PORESEG_TREE_MW=1 PORESEG_TREE_MW=0"
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; k=r["kernel_ms"]; print(d["ms_per_step"], "seq", r["sequence_ms"], "| K0", k["blocksum_ms"], "spine", k["spine_ms"], "bridge", k["bridge_ms"], "stitch", k["stitch_ms"], "tree", k["tree_ms"], "gather", k["gather_ms"], "| n", d["config"]["boundaries"], d["work"]["windows"], d["fallbacks"])'
for rep in 1 2; do
  for lib in $1; do
    [ "$lib" = "''" ] && lib=""
    export PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so
    for e in $2; do
      for s in 1 4; do echo -n "[$lib $e] streams $s: "; env ${e//,/ } python bench.py --no-cpu --no-h2d --steps 40 --warmup 8 --streams $s 2>/dev/null | python -c "$P"; done
    done
  done
done
