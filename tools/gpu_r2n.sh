#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_experiment.py -m gpu -x -q 2>&1 | tail -15
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
PORESEG_MODE=2 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_filter.py tests/test_experiment.py -m gpu -x -q 2>&1 | tail -3
