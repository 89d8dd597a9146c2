#!/bin/bash
# step time against the number of streams (contexts in flight): usage gpu_streams.sh "<lib suffix>" "<streams list>" [env...]
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "seq", r["sequence_ms"], "n", d["config"]["boundaries"])'
lib=$1; [ "$lib" = "''" ] && lib=""
export PORESEG_LIB=$PWD/pypore_amd/libporeseg$lib.so
for rep in 1 2; do for s in $2; do echo -n "[$lib ${@:3}] streams $s: "; env "${@:3}" python bench.py --no-cpu --no-h2d --no-detail --steps 60 --warmup 10 --streams $s 2>/dev/null | python -c "$P"; done; done
