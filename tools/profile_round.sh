#!/bin/bash
# Collects the evidence committed under profiles/: rocprofv3 kernel stats of the default bench command, the PMC
# passes (tools/pmc_run.sh) and the bench JSON line.  Run on the GPU box: gpurun -- 'bash tools/profile_round.sh TAG'
TAG=${1:-r01}
ROOT=$PWD
export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out
cd /tmp && rm -rf /tmp/kstats
rocprofv3 --kernel-trace --stats -d /tmp/kstats -o out --output-format csv -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu > /tmp/kstats.log 2>&1
cp $(find /tmp/kstats -name '*kernel_stats.csv' | head -1) $ROOT/gpurun_out/${TAG}_kernel_stats.csv
cd $ROOT
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
bash tools/pmc_run.sh ${TAG}_pmc > /dev/null 2>&1
head -20 gpurun_out/${TAG}_kernel_stats.csv
cat gpurun_out/${TAG}_bench.json
grep -E "^(fetch|write) " gpurun_out/${TAG}_pmc_summary.txt | grep -E "blocksum|spine|bridge|tree_"
