#!/bin/bash
# round 5: scan waves per SIMD capped by unused LDS (4 = default, 3, 2): does occupancy bound the step with 16 calls in flight?
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for pad in 0 3800 10600; do
  for ph in 1 0; do
    echo -n "[pad $pad] "; PORESEG_SCAN_LDS_PAD=$pad PORESEG_DBG_PHASE=$ph GPU_MAX_HW_QUEUES=16 python tools/bound_probe.py 16 160 2>&1 | tail -1
  done
done
done
