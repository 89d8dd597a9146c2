#!/bin/bash
# round 4: K0 at fewer waves per SIMD (it is HBM-bound; at full occupancy it keeps other calls' scan waves off the SIMDs)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "single", r["single_stream"]["sequence_ms"] if r["single_stream"] else None, {k: v for k, v in r["kernel_ms"].items() if k in ("blocksum_ms", "spine_ms", "tree_ms")})'
for rep in 1 2; do
for v in 0 4 3 2 1; do
  echo -n "[k0_waves $v] "; PORESEG_K0_WAVES=$v python bench.py --no-cpu --no-h2d --steps 100 --warmup 20 2>/dev/null | python -c "$P"
done
done
