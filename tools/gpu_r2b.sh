#!/bin/bash
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms"]; print(d["ms_per_step"], "seq", d["roofline"]["sequence_ms"], "K0", k["blocksum_ms"], "spine", k["spine_ms"], "bridge", k["bridge_ms"], "stitch", k["stitch_ms"], "tree", k["tree_ms"], "gather", k["gather_ms"], d["config"]["boundaries"], d["work"]["windows"])'
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for v in "$@"; do
  echo -n "$v : "
  env $v timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 2>gpurun_out/r2b_bench.err | python -c "$P"
done
