#!/bin/bash
# wide (64-bit) digest: new tests, filtered goldens, the filtered-event bench with and without it
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wide or beyond or dc_offset" 2>&1 | tail -15
python -m pytest tests/test_filter.py -m gpu -x -q 2>&1 | tail -5
PORESEG_MODE=2 python -m pytest tests/test_filter.py -m gpu -x -q 2>&1 | tail -5
echo "== filtered event, wide digest"; python tools/bench_filtered_parse.py 2>&1 | tail -2
echo "== filtered event, LDS-window route"; PORESEG_WIDE_BS=0 python tools/bench_filtered_parse.py 2>&1 | tail -2
