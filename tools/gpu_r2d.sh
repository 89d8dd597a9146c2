#!/bin/bash
# bench lines for library variants: usage gpu_r2d.sh "lib-suffix ENV=.." ...
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms"]; print(d["ms_per_step"], "seq", d["roofline"]["sequence_ms"], "K0", k["blocksum_ms"], "spine", k["spine_ms"], "bridge", k["bridge_ms"], "stitch", k["stitch_ms"], "tree", k["tree_ms"], "gather", k["gather_ms"], d["config"]["boundaries"], d["work"]["windows"])'
for v in "$@"; do
  set -- $v
  lib=$1; shift
  echo -n "$lib $* : "
  env PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so "$@" timeout 300 python bench.py --no-cpu --no-h2d --steps 20 --warmup 5 2>/dev/null | python -c "$P"
done
