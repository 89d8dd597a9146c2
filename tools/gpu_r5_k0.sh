#!/bin/bash
# round 5: persistent K0 (PORESEG_K0_WAVES waves per SIMD, 0 = one wave per wave block as in round 4) x shared front stream of
# the pool (PORESEG_POOL_SHARED): quick parity, then interleaved bench lines with the default 100 steps and the driver's 20
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
sha256sum pypore_amd/libporeseg.so | cut -c1-16
if [ "${SUITE:-1}" = "1" ]; then
  timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
fi
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "single", r["single_stream"]["sequence_ms"], d["config"]["checks"].get("all_streams_equal_single_stream"), d["config"]["checks"].get("g7_sha256_equal"), end=" | ")'
for rep in 1 2 ${REPS:-}; do
  for v in ${VARIANTS:-"PORESEG_K0_WAVES=0,PORESEG_POOL_SHARED=0" "PORESEG_K0_WAVES=2,PORESEG_POOL_SHARED=0" "PORESEG_K0_WAVES=0,PORESEG_POOL_SHARED=1" "PORESEG_K0_WAVES=1,PORESEG_POOL_SHARED=1" "PORESEG_K0_WAVES=2,PORESEG_POOL_SHARED=1" "PORESEG_K0_WAVES=3,PORESEG_POOL_SHARED=1" "PORESEG_K0_WAVES=4,PORESEG_POOL_SHARED=1"}; do
    echo -n "[$v] "
    env ${v//,/ } python bench.py --no-cpu --no-h2d --no-detail 2>/dev/null | python -c "$P"
    env ${v//,/ } python bench.py --no-cpu --no-h2d --no-detail --steps 20 --warmup 5 2>/dev/null | python -c "$P"
    echo
  done
done
