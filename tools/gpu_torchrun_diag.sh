#!/bin/bash
# Why is bench.py slower under torch.distributed.run?  Same box, same minute:
# plain / plain with OMP_NUM_THREADS=1 / torchrun (exports OMP_NUM_THREADS=1 when unset) / torchrun with OMP_NUM_THREADS=8.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d["roofline"]["single_stream"]; print(d["ms_per_step"], "seq", d["roofline"]["sequence_ms"], "single", s["ms_per_step"], s["sequence_ms"], d.get("host"))'
A="--no-cpu --no-h2d --no-detail --steps 100 --warmup 20"
for rep in 1 2; do
echo -n "plain: "; python bench.py $A 2>/dev/null | python -c "$P"
echo -n "plain OMP=1: "; OMP_NUM_THREADS=1 python bench.py $A 2>/dev/null | python -c "$P"
echo -n "torchrun: "; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$rep bench.py --gpus 1 $A 2>/dev/null | python -c "$P"
echo -n "torchrun OMP=8: "; OMP_NUM_THREADS=8 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2952$rep bench.py --gpus 1 $A 2>/dev/null | python -c "$P"
done
nproc; python -c "import os; print(sorted(os.sched_getaffinity(0))[:4], len(os.sched_getaffinity(0)))"
