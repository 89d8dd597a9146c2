#!/bin/bash
# round 4: the driver's command line in FRESH processes, many times, per number of contexts: does a run go wrong?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], end=" ")'
for st in ${STREAMS_LIST:-8 12 16}; do
  echo -n "[streams $st] "
  for rep in $(seq 1 ${REPS:-20}); do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-h2d --no-detail --streams $st 2>/dev/null | python -c "$P"; done; echo
done
