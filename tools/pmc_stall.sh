#!/bin/bash
# Where do the scan waves wait?  TCP / TA / UTCL1 / instruction-fetch counters of one call at a time, one pass per group
# (a group with an unknown counter name is reported and skipped).  usage: bash tools/pmc_stall.sh [extra bench args]
ROOT=$PWD; export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}     # as bench.py sets it (under rocprofv3 the runtime may start before bench.py does)
mkdir -p $ROOT/gpurun_out; cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "^\s*[A-Za-z0-9_]*\s" | sort -u > /dev/null
rocprofv3 --list-avail 2>/dev/null | grep -oE "(Name|name)[ :=]+[A-Za-z0-9_]+" | awk '{print $NF}' | sort -u | tr '\n' ' ' > $ROOT/gpurun_out/pmc_avail_names.txt
run() { name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/prof_$name -o out --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-h2d --no-detail --streams 1 $EXTRA_BENCH > /tmp/prof_$name.log 2>&1 || { echo "$name: FAILED ($*)"; tail -3 /tmp/prof_$name.log; return; }
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
for pct in 12 25 50 100; do export PORESEG_SLOTS_PCT=$pct; run occ$pct SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVES; done
# other groups used in round 3 (see profiles/r03_experiments/pmc_stall_counters.txt): TCP_UTCL1_*, TCP_TCC_READ_REQ_LATENCY_sum,
# TCP_PENDING_STALL_CYCLES_sum, SQ_IFETCH, SQC_ICACHE_*, SQ_INSTS_VALU_*_F64, SQ_INSTS_BRANCH, SQ_ACTIVE_INST_SCA ...
