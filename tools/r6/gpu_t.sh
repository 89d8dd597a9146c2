#!/bin/bash
# round 6: the profile collection again after bench.py's side measurements became medians of three (library unchanged), smoke()
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "build: $(sha256sum pypore_amd/libporeseg.so | cut -c1-64)"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
tail -3 gpurun_out/r06_profile_round.log | cut -c1-400
