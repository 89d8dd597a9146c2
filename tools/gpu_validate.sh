#!/bin/bash
# validation of a kernel change: GPU suite in default / verify / single-wave-tree / LDS-window / host-stitch modes,
# then the randomised fuzz (default + verify) and the regime parity sweep.  usage: gpu_validate.sh [fuzz seeds]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
N=${1:-3000}
# (round 4: the single-wave subtree kernel is the default, so the third mode is the multi-wave one; a sixth and seventh run
#  switch the coarse pass and the shared deep jobs off)
echo "build: $(sha256sum pypore_amd/libporeseg.so | cut -c1-64)  $(date -u +%FT%TZ)  $(python -c "from pypore_amd import _lib; print(_lib.lib().ps_version().decode())" 2>/dev/null | tail -1)"
# (round 6: the library reads no environment; tests/conftest.py turns these variables into the options every new context of the
#  test process starts with -- engine.apply_env_defaults -- so the suite below runs on the PRODUCT binary in every mode; two more
#  modes: the round-5 gather (item_scan_kernel + gather_kernel, result copy by hipMemcpyAsync), K0's 16-byte-aligned fast route, and
#  the file route by two calls instead of one pass)
for env in "X=0" "PORESEG_MODE=2" "PORESEG_TREE_MW=1" "PORESEG_SCAN_BS=0" "PORESEG_STITCH=host" "PORESEG_GROUPS=0" "PORESEG_TREE_PAR=0" \
           "PORESEG_GATHER_FUSED=0 PORESEG_DOWNLOAD=0" "PORESEG_K0_UNALIGNED=0" "PORESEG_SINGLE_PASS=0" "PORESEG_K0_WAVES=1" "PORESEG_LAT_HELP=0" "PORESEG_BRIDGE_BUDGET=3" "PORESEG_MODE=2 PORESEG_BRIDGE_BUDGET=2"; do
  echo "== $env"; env $env timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
done
FUZZ_BASE=${FUZZ_BASE:-2000000} timeout 3000 python tools/fuzz_gpu.py $N 2>&1 | tail -2 | cut -c1-420
FUZZ_SCALE=64 FUZZ_BASE=${FUZZ_BASE:-2000000} timeout 3000 python tools/fuzz_gpu.py $((N / 2)) 2>&1 | tail -3 | cut -c1-520
timeout 900 python tools/regime_parity.py 2>&1 | tail -3
timeout 600 python tools/fuzz_many_events.py 2>&1 | tail -2
# round 6: sparse shapes (helpers on / off / staying), the exact route against the oracle on random off-grid traces
timeout 900 python tools/fuzz_sparse.py $((N / 10)) 2>&1 | tail -2 | cut -c1-300
timeout 900 python tools/r6/fuzz_exact.py $((N / 20)) 2>&1 | tail -2 | cut -c1-300
timeout 900 python tools/r6/fuzz_single_pass.py $((N / 5)) 2>&1 | tail -4 | cut -c1-300
