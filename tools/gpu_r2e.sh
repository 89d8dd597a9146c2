#!/bin/bash
# library variants on several workloads: usage gpu_r2e.sh suffix...
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms"]; print(d["ms_per_step"], "seq", d["roofline"]["sequence_ms"], "K0", k["blocksum_ms"], "spine", k["spine_ms"], "bridge", k["bridge_ms"], "tree", k["tree_ms"], d["config"]["boundaries"])'
for lib in "$@"; do
  export PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so
  echo "== $lib"
  echo -n "trace 1e8: "; python bench.py --no-cpu --no-h2d --steps 20 --warmup 5 2>/dev/null | python -c "$P"
  echo -n "sharded 1e9: "; python bench.py --workload sharded-trace --steps 5 --warmup 2 2>/dev/null | python -c "$P"
  echo -n "dense dwell 150-1500: "; python bench.py --no-cpu --no-h2d --steps 5 --warmup 2 --dwell 150 1500 2>/dev/null | python -c "$P"
  echo -n "long dwell 30000-200000: "; python bench.py --no-cpu --no-h2d --steps 5 --warmup 2 --dwell 30000 200000 2>/dev/null | python -c "$P"
  echo -n "file: "; python bench.py --workload file --no-cpu --steps 20 --warmup 5 2>/dev/null | python -c "$P"
done
