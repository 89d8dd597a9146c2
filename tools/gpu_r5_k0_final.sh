cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], end=" ")'
for kw in 0 2 1; do : > /tmp/k0_$kw.txt; done
for i in 1 2 3 4 5 6; do
  for kw in 0 2 1; do
    PORESEG_K0_WAVES=$kw python bench.py --no-cpu --no-h2d --no-detail 2>/dev/null | python -c "$P" >> /tmp/k0_$kw.txt
    PORESEG_K0_WAVES=$kw python bench.py --no-cpu --no-h2d --no-detail --steps 20 --warmup 5 2>/dev/null | python -c "$P" >> /tmp/k0_$kw.txt
  done
done
for kw in 0 2 1; do echo "k0_waves $kw (100 steps, 20 steps alternating): $(cat /tmp/k0_$kw.txt)"; done
