#!/bin/bash
# round 4: the N > 1 code of bench.py (factored out this round) on the one GPU of the box with RCCL initialised -- one rank
# under torch.distributed.run with PORESEG_BENCH_DIST=1 -- for the four workloads; the trace workload also plain
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["config"]["name"], d["ms_per_step"], d["value"], "ranks_seen", d["config"]["ranks_seen"], d["config"]["ms_per_step_per_rank"], d["config"]["boundaries"], d["config"]["checks"])'
python bench.py --no-cpu --no-h2d 2>/dev/null | python -c "$P"
port=29520
for wl in "trace" "file" "sharded-trace --samples 200000000 --steps 5 --warmup 2" "files --files 6 --samples 20000000 --steps 2 --warmup 1"; do
  port=$((port + 1))
  PORESEG_BENCH_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port \
    bench.py --gpus 1 --no-cpu --no-h2d --workload $wl 2> gpurun_out/dist1_err.txt | python -c "$P" || tail -20 gpurun_out/dist1_err.txt
done
