#!/bin/bash
# careful A/B of library variants on the headline configuration: 5 rounds of (streams 4, 100 steps) per variant, interleaved
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "seq", r["sequence_ms"], "single", r["single_stream"]["sequence_ms"])'
for rep in 1 2 3 4 5; do
  for lib in "$@"; do
    [ "$lib" = "''" ] && lib=""
    echo -n "[$lib] "; PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so python bench.py --no-cpu --no-h2d --no-detail --steps 100 --warmup 20 2>/dev/null | python -c "$P"
  done
done
