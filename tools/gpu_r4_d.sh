#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_filter.py -x -q -m gpu 2>&1 | tail -15
echo "== full suite"; timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "seq", r["sequence_ms"], "single", r["single_stream"]["sequence_ms"] if r["single_stream"] else None, r["kernel_ms"])'
for s in 4 4; do
  echo -n "[streams $s] "; python bench.py --no-cpu --no-h2d --steps 120 --warmup 24 --streams $s 2>/dev/null | python -c "$P"
done
python tools/bench_filtered_parse.py 2>&1 | tail -12
