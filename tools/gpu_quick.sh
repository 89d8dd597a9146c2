#!/bin/bash
# quick GPU check used during kernel work: 1e8 digest in both scan paths + verify, then bench lines
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["work"], d["config"]["boundaries"])'
timeout 600 python tools/dbg_1e8.py 2>&1 | grep -v amdgpu.ids | tail -3
export PORESEG_SCAN_BS=1
for lib in "" noexact; do
  echo "lib=${lib:-product}"
  if [ -n "$lib" ]; then export PORESEG_LIB=$PWD/pypore_amd/libporeseg_$lib.so; fi
  timeout 300 python bench.py --no-cpu --steps 10 --warmup 2 2>/dev/null | python -c "$P"
done
export PORESEG_LIB=$PWD/pypore_amd/libporeseg_stamp.so
timeout 300 python bench.py --no-cpu --steps 1 --warmup 0 2>&1 | grep "stamps\] block" | tail -1
