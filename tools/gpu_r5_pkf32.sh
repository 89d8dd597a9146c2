#!/bin/bash
# Packed fp32 front end of K0 (PS_K0_PKF32): exactness probe of v_cvt_pknorm_i16_f32, parity, fuzz, bench A/B
# (libporeseg_pkf0.so: the same source with -DPS_K0_PKF32=0).
out=gpurun_out/r05_pkf32.txt
{
./tools/probes/pknorm_probe
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
PORESEG_K0_WAVES=1 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py -m gpu -x -q 2>&1 | tail -1
FUZZ_BASE=21000000 timeout 900 python tools/fuzz_gpu.py 1500 2>&1 | tail -1 | cut -c1-200
PORESEG_K0_WAVES=1 FUZZ_BASE=22000000 timeout 900 python tools/fuzz_gpu.py 1500 2>&1 | tail -1 | cut -c1-200
for rep in 1 2 3; do
  for lib in "" _pkf0; do
    for st in 100 20; do
      PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so python bench.py --steps $st --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('lib%-6s steps %3d  %.4f ms  frac %.4f' % ('$lib', $st, d['ms_per_step'], d['roofline']['frac']))"
    done
  done
done
} > $out 2>&1
cat $out
