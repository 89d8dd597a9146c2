#!/bin/bash
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["streams"], "ms/step", d["ms_per_step"], "frac", r["frac"], "single", r["single_stream"]["ms_per_step"] if r["single_stream"] else None, d["config"]["boundaries"])'
for lib in "$@"; do
  export PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so
  for rep in 1 2; do echo -n "$lib trace: "; python bench.py --no-cpu --no-h2d --no-detail 2>/dev/null | python -c "$P"; done
  echo -n "$lib file : "; python bench.py --workload file --no-cpu --no-detail 2>/dev/null | python -c "$P"
done
