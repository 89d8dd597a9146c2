#!/bin/bash
# round 4: wider range of the expansion behind the group bound (t <= 0.3, c(t) = 0.5 + 0.44 t) and the K0 trims:
# audit + suite of the new default, then interleaved A/B: _base (build 0d60ef1b), _s (range only), "" (range + K0 trims)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
sha256sum pypore_amd/libporeseg*.so | cut -c1-16,65-
timeout 900 python -m pytest tests/test_bound_audit.py -x -q -m gpu 2>&1 | tail -4
for env in "X=0" "PORESEG_MODE=2"; do
  echo "== new $env"; env $env timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
done
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "single", r["single_stream"]["sequence_ms"], {k: v for k, v in r["kernel_ms"].items() if k in ("blocksum_ms", "spine_ms", "tree_ms")}, d["work"].get("rows_per_window"))'
for rep in 1 2 3 4 5; do
  for lib in _base _s ""; do
    echo -n "[$lib] "; PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so python bench.py --no-cpu --no-h2d --steps 100 --warmup 20 2>/dev/null | python -c "$P"
  done
done
bash tools/pmc_run.sh r4slack_pmc 1 2>&1 | grep -E "^sq (void )?ps::(blocksum|spine|tree)" | cut -c1-260
