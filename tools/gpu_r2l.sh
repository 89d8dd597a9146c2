#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
PORESEG_MODE=2 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_filter.py -m gpu -x -q 2>&1 | tail -3
for s in 1 4 4; do echo -n "streams $s: "; python bench.py --no-cpu --no-h2d --steps 20 --warmup 5 --streams $s 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"])'; done
