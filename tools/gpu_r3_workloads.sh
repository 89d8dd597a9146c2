#!/bin/bash
# every bench workload and the side benches on the round-3 build (numbers for DESIGN.md 7 / 7b / 7c)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["config"]["name"], d["ms_per_step"], "ms/step", d["value"], d["unit"], "frac", d["roofline"]["frac"], d["config"]["boundaries"][:2], d["config"]["checks"], d["fallbacks"])'
for wl in file sharded-trace; do timeout 900 python bench.py --no-cpu --workload $wl 2>gpurun_out/r3_wl_$wl.err | python -c "$P"; done
timeout 900 python bench.py --no-cpu --workload sharded-trace --dwell 100000 1000000 --steps 5 --warmup 2 2>>gpurun_out/r3_wl_sharded-trace.err | python -c "$P"
timeout 900 python bench.py --no-cpu --workload files --steps 2 --warmup 1 2>gpurun_out/r3_wl_files.err | python -c "$P"
timeout 600 python tools/bench_config2.py 2>&1 | tail -3
timeout 600 python tools/bench_filtered_parse.py 2>&1 | tail -8
timeout 600 python tools/bench_filter.py 2>&1 | tail -4
timeout 900 python tools/bench_experiment.py 2>&1 | tail -4
timeout 600 python tools/bench_align.py 2>&1 | tail -4
