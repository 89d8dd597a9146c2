#!/bin/bash
# round 6: raised issue priority for the pipeline's small kernels (PS_SMALL_PRIO 2 / 3 against 0): interleaved bench runs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], "seq", r["sequence_ms"])'
for rep in 1 2 3 4 5 6; do
  for lib in _prio0 _prio2 _prio3; do
    echo -n "[$lib] 100: "; PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so python bench.py --no-cpu --no-h2d --no-detail --steps 100 --warmup 20 2>/dev/null | python -c "$P"
    echo -n "[$lib]  20: "; PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so python bench.py --no-cpu --no-h2d --no-detail --steps 20 --warmup 5 2>/dev/null | python -c "$P"
  done
done 2>&1 | tee gpurun_out/r6_small_prio_ab.txt
