#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for s in 1 4 4 3 6; do echo -n "streams $s: "; python bench.py --no-cpu --no-h2d --steps 40 --warmup 8 --streams $s 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"])'; done
echo -n "file: "; python bench.py --workload file --no-cpu --steps 20 --warmup 5 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms"])'
echo -n "sharded 1e9: "; python bench.py --workload sharded-trace --steps 5 --warmup 2 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms"])'
