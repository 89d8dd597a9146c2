#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for env in "X=0" "PORESEG_GATHER_FUSED=0"; do echo "== $env"; env $env python tools/bench_config2.py 2>&1 | tail -2 | cut -c1-230; done
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
