#!/bin/bash
# rocprofv3 counter passes for bench.py (run on the GPU box through gpurun); summaries land in gpurun_out/pmc_*.txt
# usage: bash tools/pmc_run.sh [tag] [streams]   (extra env such as PORESEG_LIB is inherited)
TAG=${1:-pmc}
STREAMS=${2:-1}
ROOT=$PWD
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}     # as bench.py sets it (under rocprofv3 the runtime may start before bench.py does)
mkdir -p $ROOT/gpurun_out
cd /tmp
run() {  # name counters...
  name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --pmc "$@" --kernel-trace -d /tmp/prof_$name -o out --output-format csv -- python3 $ROOT/bench.py --steps $((3 * STREAMS)) --warmup $STREAMS --no-cpu --no-h2d --no-detail --streams $STREAMS > /tmp/prof_$name.log 2>&1
  python3 - "$name" <<'PY' >> $ROOT/gpurun_out/${TAG}_summary.txt
import sys, csv, glob, collections
name = sys.argv[1]
files = glob.glob('/tmp/prof_%s/**/*counter_collection.csv' % name, recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'ps::' not in k: continue
        k = k.split('(')[0]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
for k in sorted(acc):
    print(name, k, {c: int(v / cnt[(k, c)]) for c, v in sorted(acc[k].items())})
PY
}
: > $ROOT/gpurun_out/${TAG}_summary.txt
run sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
cat $ROOT/gpurun_out/${TAG}_summary.txt
