#!/bin/bash
# round 4: where a window's vector instructions go.  Library variants that leave the scan after its setup (cut1), after the
# coarse pass (cut2), after the sweep (cut3), after the drain (cut4) -- WRONG results, instruction counts only -- and the
# product library, one SQ counter pass each (one call at a time); prints instructions per window of the scan kernels.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8 TMPDIR=/tmp
ROOT=$PWD
cd /tmp
for v in "" _cut1 _cut2 _cut3 _cut4; do
  rm -rf /tmp/prof_cut
  PORESEG_LIB=$ROOT/pypore_amd/libporeseg$v.so PORESEG_BENCH_NOCHECK=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace -d /tmp/prof_cut -o out --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-h2d --no-detail --streams 1 > /tmp/prof_cut.log 2>&1
  tail -1 /tmp/prof_cut.log | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('[$v] work', d['work'])
except Exception as e:
    print('[$v] no bench line', e)"
  python3 - "$v" <<'PY'
import sys, csv, glob, collections
files = glob.glob('/tmp/prof_cut/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if not any(s in k for s in ('spine_kernel', 'tree_kernel', 'bridge_kernel', 'blocksum')): continue
        k = k.split('(')[0].replace('void ps::', '')
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k in sorted(acc):
    print('   ', sys.argv[1] or 'product', k, {c: int(v / cnt[(k, c)]) for c, v in sorted(acc[k].items())})
PY
done
