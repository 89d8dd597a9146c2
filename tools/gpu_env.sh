#!/bin/bash
# bench lines for a list of env settings: usage gpu_env.sh "A=1 B=2" "A=3" ...
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms"]; print(d["ms_per_step"], "K0", k["blocksum_ms"], "spine", k["spine_ms"], "bridge", k["bridge_ms"], "tree", k["tree_ms"], "gather", k["gather_ms"], d["config"]["boundaries"])'
for v in "$@"; do
  echo -n "$v : "
  env $v timeout 300 python bench.py --no-cpu --steps 10 --warmup 2 2>/dev/null | python -c "$P"
done
