#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wide or beyond or dc_offset" 2>&1 | tail -2
python -m pytest tests/test_filter.py -m gpu -x -q 2>&1 | tail -2
for t in 0 40000 24000 16000 12000 8000; do echo -n "tile $t: "; TILE=$t python tools/bench_filtered_parse.py 2>&1 | tail -1 | cut -c1-330; done
