#!/bin/bash
# round 5: the driver's command line in fresh processes, K0 admission 3 / off alternating: median, worst, runs gone wrong
cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'
for adm in 3 0; do : > /tmp/fresh_$adm.txt; done
for i in $(seq 1 ${N:-24}); do
  for adm in 3 0; do
    PORESEG_POOL_K0_MAX=$adm python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-h2d --no-detail 2>/dev/null | python -c "$P" >> /tmp/fresh_$adm.txt
  done
done
for adm in 3 0; do
  python - $adm <<'PY'
import sys
v = sorted(float(x) for x in open('/tmp/fresh_%s.txt' % sys.argv[1]))
print("admit %s: n %d median %.4f min %.4f max %.4f above 0.25: %d  all: %s" % (sys.argv[1], len(v), v[len(v)//2], v[0], v[-1], sum(x > 0.25 for x in v), " ".join("%.3f" % x for x in v)))
PY
done
