#!/bin/bash
# round 4, final build: step time against the number of contexts in flight and of hardware queues, interleaved rounds;
# "short": only the driver's own command line (--steps 20 --warmup 5)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], end=" ")'
if [ "$1" != short ]; then
for rep in 1 2 3 4 5 6 7 8; do
  for v in "4 8" "8 8" "8 16" "12 16" "16 16" "12 24" "16 24" "24 24"; do
    set -- $v
    echo -n "[streams $1 queues $2] "; GPU_MAX_HW_QUEUES=$2 python bench.py --no-cpu --no-h2d --no-detail --steps 240 --warmup 48 --streams $1 2>/dev/null | python -c "$P"
  done; echo
done
fi
echo "== --steps 20 --warmup 5"
for rep in 1 2 3 4 5 6 7 8; do
  for v in "4 16" "5 16" "6 16" "7 16" "8 16" "10 16" "12 16"; do
    set -- $v
    echo -n "[streams $1 queues $2] "; GPU_MAX_HW_QUEUES=$2 python bench.py --no-cpu --no-h2d --no-detail --steps 20 --warmup 5 --streams $1 2>/dev/null | python -c "$P"
  done; echo
done
