#!/bin/bash
# subtrees by levels: whole GPU suite with it forced on, then the filtered event for several level counts
mkdir -p gpurun_out
echo "== suite, PORESEG_TREE_LEVELS=3"; PORESEG_TREE_LEVELS=3 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== parity, PORESEG_TREE_LEVELS=1"; PORESEG_TREE_LEVELS=1 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py -m gpu -x -q 2>&1 | tail -3
echo "== parity, PORESEG_TREE_LEVELS=12 verify mode"; PORESEG_MODE=2 PORESEG_TREE_LEVELS=12 python -m pytest tests/test_gpu_parity.py tests/test_filter.py -m gpu -x -q 2>&1 | tail -3
echo "== default suite"; python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for lv in 0 1 2 4 6 8 10 14; do echo -n "levels $lv: "; PORESEG_TREE_LEVELS=$lv python tools/bench_filtered_parse.py 2>&1 | tail -1 | cut -c1-260; done
