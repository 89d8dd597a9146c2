#!/bin/bash
# round 5: s_setprio on K0's waves (library variants _p1, _p3), whole call and K0 alone, unshared and on the shared front stream
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for lib in "" _p1 _p3; do
  export PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so
  for cfg in "0 0 2" "0 0 1" "0 0 3" "0 2 2" "1 0 2"; do
    set -- $cfg
    echo -n "[lib$lib] "; PORESEG_POOL_SHARED=$1 PORESEG_DBG_PHASE=$2 PORESEG_K0_WAVES=$3 python tools/bound_probe.py 16 160 2>&1 | tail -1
  done
done
done
