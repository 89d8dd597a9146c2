#!/bin/bash
# round 6, item 5: the single-pass file route -- its tests, then the bench's int16_file block (single pass against the two calls)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_single_pass.py tests/test_full_size.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15
for i in 1 2; do
timeout 600 python bench.py --no-cpu --no-h2d 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print(d['ms_per_step'], json.dumps(d.get('int16_file'))[:900])"
done
