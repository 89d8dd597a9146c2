#!/bin/bash
# average latency of the vector / scalar / LDS memory instructions of the scan kernels: SQ_INST_LEVEL_x / SQ_INSTS_x
ROOT=$PWD; export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}     # as bench.py sets it (under rocprofv3 the runtime may start before bench.py does)
mkdir -p $ROOT/gpurun_out; cd /tmp
run() { name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --pmc "$@" --kernel-trace -d /tmp/prof_$name -o out --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-h2d --no-detail --streams 1 > /tmp/prof_$name.log 2>&1
  python3 - "$name" <<'PY'
import sys, csv, glob, collections
name = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('/tmp/prof_%s/**/*counter_collection.csv' % name, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'ps::' not in k or 'synth' in k: continue
        k = k.split('(')[0]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k in sorted(acc):
    if any(x in k for x in ('spine', 'tree', 'blocksum', 'bridge_kernel')):
        print(name, k, {c: int(v / cnt[(k, c)]) for c, v in sorted(acc[k].items())})
PY
}
run l1 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES
run l2 SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS
run l3 SQ_INSTS_FLAT SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM
