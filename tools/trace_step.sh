#!/bin/bash
# kernel timeline of the last bench step (rocprofv3 --kernel-trace): start offset and duration of every kernel
ROOT=$PWD
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/ktrace
rocprofv3 --kernel-trace -d /tmp/ktrace -o out --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-detail "$@" > /tmp/ktrace.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/ktrace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# last occurrence of blocksum kernel = start of last step
idx = max(i for i, r in enumerate(rows) if 'blocksum' in r['Kernel_Name'] or 'spine_kernel' in r['Kernel_Name'] and not any('blocksum' in x['Kernel_Name'] for x in rows))
t0 = int(rows[idx]['Start_Timestamp'])
prev_end = t0
for r in rows[idx:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%8.1f us  +gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r['Kernel_Name'].split('(')[0][:60]))
    prev_end = e
PY
