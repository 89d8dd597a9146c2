#!/bin/bash
# Kernel timeline of the default bench (4 streams): how much of the time two "full-chip" kernels (spine / subtree) of
# different calls run at the same time.  Run on the GPU box; prints a summary.
ROOT=$PWD; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/ktrace
rocprofv3 --kernel-trace -d /tmp/ktrace -o out --output-format csv -- python3 $ROOT/bench.py --steps 40 --warmup 5 --no-cpu --no-h2d --no-detail "$@" > /tmp/ktrace.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/ktrace/**/*kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    kind = 'spine' if 'spine_kernel' in n else 'tree' if 'tree_mw' in n or 'tree_kernel' in n else 'k0' if 'blocksum' in n else 'bridge' if 'bridge' in n else 'other'
    if 'synth' in n: continue
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), kind, r.get('Queue_Id', r.get('Stream_Id', '?'))))
rows.sort()
t0 = rows[len(rows) // 3][0]; t1 = rows[-len(rows) // 6][1]       # steady state of the timed region
rows = [r for r in rows if r[0] >= t0 and r[1] <= t1]
span = (t1 - t0) / 1e3
# sweep line: time with k big kernels (spine / tree) active, and with any kernel active
ev = []
for s, e, k, q in rows:
    ev.append((s, 1, k)); ev.append((e, -1, k))
ev.sort()
big = 0; anyk = 0; last = ev[0][0]; tb = collections.Counter(); ta = collections.Counter()
for t, d, k in ev:
    tb[big] += t - last; ta[anyk] += t - last; last = t
    anyk += d
    if k in ('spine', 'tree'): big += d
print("steady window %.1f us, kernels %d" % (span, len(rows)))
print("time with n spine/subtree kernels active:", {n: "%.1f%%" % (100.0 * v / (t1 - t0)) for n, v in sorted(tb.items())})
print("time with n kernels of any kind active:", {n: "%.1f%%" % (100.0 * v / (t1 - t0)) for n, v in sorted(ta.items())})
dur = collections.defaultdict(list)
for s, e, k, q in rows: dur[k].append((e - s) / 1e3)
print({k: "%.1f us x %d" % (sum(v) / len(v), len(v)) for k, v in dur.items()})
PY
