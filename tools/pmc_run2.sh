#!/bin/bash
# second set of counter passes (instruction cache, LDS, issue stalls); see tools/pmc_run.sh
TAG=${1:-pmc2}
ROOT=$PWD
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}     # as bench.py sets it (under rocprofv3 the runtime may start before bench.py does)
mkdir -p $ROOT/gpurun_out
cd /tmp
run() {
  name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --pmc "$@" --kernel-trace -d /tmp/prof_$name -o out --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu > /tmp/prof_$name.log 2>&1
  tail -3 /tmp/prof_$name.log | grep -i "error\|invalid\|not" | head -3
  python3 - "$name" <<'PY' >> $ROOT/gpurun_out/${TAG}_summary.txt
import sys, csv, glob, collections
name = sys.argv[1]
files = glob.glob('/tmp/prof_%s/**/*counter_collection.csv' % name, recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'spine' not in k and 'tree_' not in k and 'bridge' not in k: continue
        k = k.split('(')[0]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
for k in sorted(acc):
    print(name, k, {c: int(v / cnt[(k, c)]) for c, v in sorted(acc[k].items())})
PY
}
: > $ROOT/gpurun_out/${TAG}_summary.txt
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES
run act SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES
run dc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_WAVE_CYCLES
cat $ROOT/gpurun_out/${TAG}_summary.txt
