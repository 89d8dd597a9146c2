#!/bin/bash
# Collects the evidence committed under profiles/ for one round.  Run on the GPU box:
#   gpurun -- 'bash tools/profile_round.sh r02'
# Outputs (gpurun_out/, copy to profiles/):
#   <tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the DEFAULT bench command (sixteen contexts in flight: kernels
#                               of different calls overlap, so single launches run longer than alone)
#   <tag>_s1_kernel_stats.csv   the same with --streams 1 (one call at a time: the per-kernel durations of roofline.kernel_ms)
#   <tag>_bench.json            the default bench line (incl. cpu_baseline, h2d_inclusive)
#   <tag>_pmc_summary.txt       PMC passes (tools/pmc_run.sh, --streams 1: instruction counts, traffic of one call at a time)
#   <tag>_pmc16_summary.txt     the same passes with sixteen calls in flight on sixteen DISTINCT traces (--streams 16, the bench's
#                               default): what the headline configuration fetches; <tag>_pmc_traffic.json: HBM bytes per launch from both
TAG=${1:-r06}
ROOT=$PWD
export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out
# (bench.py sets this itself, but under rocprofv3 the tool library may start the runtime first: the profiles and the headline
#  must run on the same number of hardware queues -- ADVICE r3)
export GPU_MAX_HW_QUEUES=16
echo "GPU_MAX_HW_QUEUES=$GPU_MAX_HW_QUEUES" > $ROOT/gpurun_out/${TAG}_profile_env.txt
cd /tmp
for s in 16 1; do
  rm -rf /tmp/kstats
  # (--no-detail since round 6: without it the line's side workloads -- config 2, config 3 -- launch the same kernels on other
  #  shapes and their durations are averaged into the headline's: K0 of a 5.12e7-sample batch takes half as long)
  rocprofv3 --kernel-trace --stats -d /tmp/kstats -o out --output-format csv -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-h2d --no-detail --streams $s > /tmp/kstats_$s.log 2>&1
  out=$ROOT/gpurun_out/${TAG}_kernel_stats.csv
  [ $s = 1 ] && out=$ROOT/gpurun_out/${TAG}_s1_kernel_stats.csv
  cp $(find /tmp/kstats -name '*kernel_stats.csv' | head -1) $out
  tail -1 /tmp/kstats_$s.log | cut -c1-400
done
cd $ROOT
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_20.json 2>> gpurun_out/${TAG}_bench.err
bash tools/pmc_run.sh ${TAG}_pmc 1 > /dev/null 2>&1
bash tools/pmc_run.sh ${TAG}_pmc16 16 > /dev/null 2>&1
python3 - $TAG <<'PY'
import sys, re, json, ast
tag = sys.argv[1]
def read(name):
    per = {}
    for line in open('gpurun_out/%s_summary.txt' % name):
        m = re.match(r'(fetch|write|sq) (?:void )?ps::(\w+)(?:<[^>]*>)? (\{.*\})', line.strip())
        if not m: continue
        d = ast.literal_eval(m.group(3))
        k = m.group(2)
        per.setdefault(k, {})
        if m.group(1) == 'fetch': per[k]['fetch_kib'] = d['FETCH_SIZE']
        elif m.group(1) == 'write': per[k]['write_kib'] = d['WRITE_SIZE']
        elif 'SQ_INSTS_VALU' in d: per[k]['valu'] = d['SQ_INSTS_VALU']
    return per
per4 = read(tag + '_pmc16')
per = {}
for line in open('gpurun_out/%s_pmc_summary.txt' % tag):
    m = re.match(r'(fetch|write|sq) (?:void )?ps::(\w+)(?:<[^>]*>)? (\{.*\})', line.strip())
    if not m: continue
    d = ast.literal_eval(m.group(3))
    k = m.group(2)
    per.setdefault(k, {})
    if m.group(1) == 'fetch': per[k]['fetch_kib'] = d['FETCH_SIZE']
    elif m.group(1) == 'write': per[k]['write_kib'] = d['WRITE_SIZE']
    elif 'SQ_INSTS_VALU' in d: per[k]['valu'] = d['SQ_INSTS_VALU']
names = ['blocksum_kernel', 'spine_kernel', 'bridge_kernel', 'bridge_la_kernel', 'tree_kernel', 'tree_mw_kernel', 'assemble_tiles_kernel',
         'assemble_items_kernel', 'item_scan_kernel', 'gather_kernel', 'gather_scan_kernel', 'download_kernel', 'upload_kernel']
pk = {k: (2 * per[k].get('fetch_kib', 0) + per[k].get('write_kib', 0)) * 1024 for k in names if k in per}
pk4 = {k: (2 * per4[k].get('fetch_kib', 0) + per4[k].get('write_kib', 0)) * 1024 for k in names if k in per4}
json.dump({"source": "profiles/%s_pmc_summary.txt: FETCH_SIZE (KiB) x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE (KiB) per "
                     "launch, rocprofv3 --pmc passes of `bench.py --steps 3 --warmup 1 --no-cpu --no-h2d --streams 1`" % tag,
           "total": sum(pk.values()), "per_kernel": pk,
           "in_flight_streams": 16, "total_in_flight": sum(pk4.values()), "per_kernel_in_flight": pk4,
           "source_in_flight": "profiles/%s_pmc16_summary.txt: the same passes with --streams 16 (sixteen calls in flight, one trace each)" % tag,
           "valu_source": "profiles/%s_pmc_summary.txt: SQ_INSTS_VALU (wave-level vector instructions) per launch, same passes" % tag,
           "valu_per_kernel": {k: per[k]['valu'] for k in names if k in per and 'valu' in per[k]}},
          open('gpurun_out/%s_pmc_traffic.json' % tag, 'w'), indent=1)
print(json.dumps(pk))
PY
head -12 gpurun_out/${TAG}_s1_kernel_stats.csv | cut -c1-60,200-400
cat gpurun_out/${TAG}_bench.json | cut -c1-600
