#!/bin/bash
# round 5: scans alone / K0 alone / whole call, T calls in flight (tools/bound_probe.py)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export PORESEG_POOL_SHARED=${PORESEG_POOL_SHARED:-0}
for T in ${TS:-16 8 4 1}; do
  for ph in 0 1 2; do
    for kw in ${KWS:-2 0}; do
      PORESEG_DBG_PHASE=$ph PORESEG_K0_WAVES=$kw python tools/bound_probe.py $T ${K:-160} 2>&1 | tail -1
    done
  done
done
