#!/bin/bash
# the nominal bench with the helpers on / off (lone-call figures are the ones to watch), then the sparse regimes
out=gpurun_out/r05_lat_help2.txt
{
for rep in 1 2 3 4; do
  for on in 1 0; do
      PORESEG_LAT_HELP=$on python bench.py --steps 60 --warmup 5 --no-cpu --no-h2d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('lat_help $on  %.4f ms  single %.4f  int16 %.4f  config2 %.4f (seq %.4f) / %.4f' % (d['ms_per_step'], d['roofline']['single_stream']['sequence_ms'], d['int16_file']['ms_per_step'], d['config2']['ms_per_step'], d['config2']['sequence_ms'], d['config2']['in_flight']['ms_per_step']))"
  done
done
timeout 600 python tools/dbg_lat_help.py 2>&1 | grep -v amdgpu.ids | tail -22
} > $out 2>&1
cat $out
