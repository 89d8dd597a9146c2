#!/bin/bash
# kernels AND memory copies of the last bench step on one timeline (rocprofv3 --kernel-trace --memory-copy-trace)
ROOT=$PWD
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/ctrace
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/ctrace -o out --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-detail "$@" > /tmp/ctrace.log 2>&1
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob('/tmp/ctrace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:50]))
for f in glob.glob('/tmp/ctrace/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY %s %s B' % (r.get('Direction', '?'), r.get('Bytes', r.get('Size', '?')))))
ev.sort()
last = max(i for i, e in enumerate(ev) if 'blocksum' in e[2])
i0 = last
while i0 > 0 and ev[last][0] - ev[i0 - 1][1] < 60000 and 'gather' not in ev[i0 - 1][2]: i0 -= 1
t0 = ev[i0][0]; prev = t0
for s, e, n in ev[i0:]:
    print("%8.1f us  +gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n)); prev = e
PY
