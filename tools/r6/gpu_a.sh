#!/bin/bash
# round 6, first GPU call: the suite on the new build (no getenv in the product, lat_help 2, shared_device, gate), baseline bench
# lines on this box, the K0 group-record A/B (next #1a), the config-2 tile probe (next #6)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "build: $(sha256sum pypore_amd/libporeseg.so | cut -c1-16) diag $(sha256sum pypore_amd/libporeseg_diag.so | cut -c1-16)  $(date -u +%FT%TZ)"
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["sequence_ms"], (d["roofline"]["single_stream"] or {}).get("sequence_ms"), d.get("int16_file",{}).get("ms_per_step"), d.get("config2",{}).get("ms_per_step"), d.get("config2",{}).get("in_flight",{}).get("ms_per_step"))'
for i in 1 2 3; do
  python bench.py --no-cpu --no-h2d --steps 20 --warmup 5 2>/dev/null | python -c "$P"
done
python bench.py --no-cpu --no-h2d 2>/dev/null | python -c "$P"
PORESEG_LIB=$PWD/pypore_amd/libporeseg_diag.so timeout 900 python tools/r6/k0_grp_probe.py 16 100 12 2>&1 | tail -8 | tee gpurun_out/r6_k0_grp_probe.txt
timeout 600 python tools/r6/config2_tile_probe.py 16 2>&1 | tail -14 | tee gpurun_out/r6_config2_tile.txt
