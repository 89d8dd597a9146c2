#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
python bench.py --no-cpu --no-h2d --steps 20 --warmup 5 --streams 1 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["work"], d["roofline"]["kernel_ms"])'
export TMPDIR=/tmp; ROOT=$PWD; cd /tmp
rm -rf /tmp/prof_v; rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace -d /tmp/prof_v -o out --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-h2d --streams 1 > /tmp/prof_v.log 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('/tmp/prof_v/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'ps::' not in k: continue
        k = k.split('(')[0]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k in sorted(acc): print(k, {c: int(v / cnt[(k, c)]) for c, v in sorted(acc[k].items())})
PY
