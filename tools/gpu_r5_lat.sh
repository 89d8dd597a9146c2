#!/bin/bash
# helpers of the look-ahead kernel (lat_help): parity, fuzz, regimes, and the nominal bench with the helpers on / off
out=gpurun_out/r05_lat_help.txt
{
python -m pytest tests -x -q -m gpu 2>&1 | tail -1
PORESEG_MODE=2 python -m pytest tests -x -q -m gpu 2>&1 | tail -1
timeout 600 python tools/dbg_lat_help.py 2>&1 | grep -v amdgpu.ids
FUZZ_BASE=51000000 timeout 900 python tools/fuzz_gpu.py 1500 2>&1 | tail -1 | cut -c1-200
timeout 600 python tools/regime_parity.py 2>&1 | tail -2
for rep in 1 2 3; do
  for on in 1 0; do
    for st in 100 20; do
      PORESEG_LAT_HELP=$on python bench.py --steps $st --warmup 5 --no-cpu --no-h2d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('lat_help $on steps %3d  %.4f ms  single %.4f  int16 %.4f  config2 %.4f / %.4f' % ($st, d['ms_per_step'], d['roofline']['single_stream']['sequence_ms'], d['int16_file']['ms_per_step'], d['config2']['ms_per_step'], d['config2']['in_flight']['ms_per_step']))"
    done
  done
done
} > $out 2>&1
cat $out
