#!/bin/bash
# A/B of library variants on the bench trace: usage gpu_ab.sh suffix... ("" = the product library), three rounds
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms"]; print(d["ms_per_step"], "spine", k["spine_ms"], "bridge", k["bridge_ms"], "tree", k["tree_ms"], "K0", k["blocksum_ms"])'
for rep in 1 2 3; do
  for lib in "$@"; do
    export PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so
    for s in 1 4; do echo -n "[$lib] streams $s: "; python bench.py --no-cpu --no-h2d --steps 40 --warmup 8 --streams $s 2>/dev/null | python -c "$P"; done
  done
done
