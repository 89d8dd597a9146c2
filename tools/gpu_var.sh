#!/bin/bash
# bench lines for library variants: usage gpu_var.sh name[:TILE] ...
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms"]; print(d["ms_per_step"], "K0", k["blocksum_ms"], "spine", k["spine_ms"], "bridge", k["bridge_ms"], "tree", k["tree_ms"], "tiles", d["work"]["tiles"], d["config"]["boundaries"])'
for v in "$@"; do
  lib=${v%%:*}; tile=${v#*:}; [ "$tile" = "$v" ] && tile=0
  echo -n "$lib tile=$tile : "
  PORESEG_TILE=$tile PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so timeout 300 python bench.py --no-cpu --steps 10 --warmup 2 2>/dev/null | python -c "$P"
done
