#!/bin/bash
# round 4: K0 reading the samples with non-temporal loads (variant _nt), five interleaved rounds + the 1e9 trace
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "single", r["single_stream"]["sequence_ms"] if r["single_stream"] else None, r["kernel_ms"]["blocksum_ms"])'
for rep in 1 2 3 4 5; do
  for lib in "" _nt; do
    echo -n "[$lib] "; PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so python bench.py --no-cpu --no-h2d --steps 100 --warmup 20 2>/dev/null | python -c "$P"
  done
done
for lib in "" _nt ""  _nt; do
  echo -n "[1e9 $lib] "; PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so python bench.py --no-cpu --workload sharded-trace --steps 10 --warmup 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms"]["blocksum_ms"])'
done
