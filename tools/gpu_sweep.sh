#!/bin/bash
# sweep: occupancy variants x tile counts for the block-sum scan (bench lines only)
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms"]; print(d["ms_per_step"], k["blocksum_ms"], k["spine_ms"], k["bridge_ms"], k["tree_ms"], d["work"]["tiles"], d["work"]["windows"], d["config"]["boundaries"])'
for lib in "" _w3 _w4; do
  for tile in 48832 32552 24416 16280; do
    echo -n "lib=libporeseg$lib tile=$tile : "
    PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so PORESEG_TILE=$tile timeout 300 python bench.py --no-cpu --steps 10 --warmup 2 2>/dev/null | python -c "$P"
  done
done
