#!/bin/bash
# long randomised soak of the final build (fresh seeds): narrow and scale-64 fuzz in default + verify mode, many short
# events, the filter fuzz, the regime sweep
mkdir -p gpurun_out
FUZZ_BASE=${FUZZ_BASE:-9000000}
FUZZ_BASE=$FUZZ_BASE timeout 2400 python tools/fuzz_gpu.py ${1:-8000} 2>&1 | tail -2 | cut -c1-300
FUZZ_SCALE=64 FUZZ_BASE=$FUZZ_BASE timeout 1200 python tools/fuzz_gpu.py ${2:-6000} 2>&1 | tail -2 | cut -c1-300
FUZZ_SCALE=16 FUZZ_BASE=$((FUZZ_BASE + 500000)) timeout 1200 python tools/fuzz_gpu.py ${2:-6000} 2>&1 | tail -2 | cut -c1-300
timeout 900 python tools/fuzz_many_events.py 2>&1 | tail -2 | cut -c1-300
timeout 600 python tools/fuzz_filter.py 2>&1 | tail -1
timeout 900 python tools/regime_parity.py 2>&1 | tail -2
