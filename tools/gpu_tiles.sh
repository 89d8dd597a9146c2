#!/bin/bash
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms"]; t=d["work"]["tiles"]; print(d["ms_per_step"], "spine", k["spine_ms"], "bridge", k["bridge_ms"], "tree", k["tree_ms"], "tiles", t, "us/window(spine)", round(1000*k["spine_ms"]/(19239.0/t),2))'
export PORESEG_SCAN_BS=1
for tile in 1562500 781250 390625 195313 97657 48832; do
  echo -n "tile=$tile : "
  PORESEG_TILE=$tile timeout 300 python bench.py --no-cpu --steps 5 --warmup 1 2>/dev/null | python -c "$P"
done
