#!/bin/bash
# instruction-fetch counters of the scan kernels (one call at a time): is the 84 KB spine kernel thrashing the I-cache?
ROOT=$PWD; export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}     # as bench.py sets it (under rocprofv3 the runtime may start before bench.py does)
mkdir -p $ROOT/gpurun_out; cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -io "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQC_TC[A-Z_]*\|SQC_DCACHE[A-Z_]*\|SQ_WAVE_CYCLES\|SQ_BUSY_CU_CYCLES\|SQ_VALU_MFMA_BUSY_CYCLES\|SQ_ACTIVE_INST_[A-Z_]*\|SQ_INST_CYCLES_[A-Z_]*\|SQ_THREAD_CYCLES_VALU\|SQ_WAIT_INST_LDS\|SQ_LDS_[A-Z_]*" | sort -u | tr '\n' ' ' > $ROOT/gpurun_out/pmc_avail.txt
echo >> $ROOT/gpurun_out/pmc_avail.txt
run() { name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --pmc "$@" --kernel-trace -d /tmp/prof_$name -o out --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-h2d --streams 1 > /tmp/prof_$name.log 2>&1
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
    if any(x in k for x in ('spine', 'tree_mw', 'blocksum', 'bridge_kernel')):
        print(name, k, {c: int(v / cnt[(k, c)]) for c, v in sorted(acc[k].items())})
PY
}
run ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run ic2 SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run ic3 SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
run ic4 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQC_DCACHE_REQ SQC_DCACHE_MISSES
