cd "$GRAFT_REPO_ROOT"
for sh in 0 1; do for ph in 2 0; do for kw in 2 0; do
PORESEG_POOL_SHARED=$sh PORESEG_DBG_PHASE=$ph PORESEG_K0_WAVES=$kw python tools/bound_probe.py 16 160 2>&1 | tail -1
done; done; done
PORESEG_POOL_SHARED=1 PORESEG_FRONT_PRIORITY=1 PORESEG_DBG_PHASE=0 PORESEG_K0_WAVES=2 python tools/bound_probe.py 16 160 2>&1 | tail -1
PORESEG_POOL_SHARED=1 PORESEG_FRONT_PRIORITY=1 PORESEG_DBG_PHASE=0 PORESEG_K0_WAVES=4 python tools/bound_probe.py 16 160 2>&1 | tail -1
