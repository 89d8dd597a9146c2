#!/bin/bash
# round 2, run A: parity suite on the new build, then bench lines with and without the speculative subtree queue
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms"]; print(d["ms_per_step"], "seq", d["roofline"]["sequence_ms"], "K0", k["blocksum_ms"], "spine", k["spine_ms"], "bridge", k["bridge_ms"], "stitch", k["stitch_ms"], "tree", k["tree_ms"], "gather", k["gather_ms"], d["config"]["boundaries"], d["work"])'
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2a_pytest.log
cat gpurun_out/r2a_pytest.log
for v in "PORESEG_SPEC_TREE=1" "PORESEG_SPEC_TREE=0"; do
  echo -n "$v : "
  env $v timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 2>gpurun_out/r2a_bench.err | python -c "$P"
done
tail -5 gpurun_out/r2a_bench.err
