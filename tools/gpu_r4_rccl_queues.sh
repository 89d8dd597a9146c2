cd "$GRAFT_REPO_ROOT"
export MASTER_ADDR=127.0.0.1
for rep in 1 2 3; do
for q in 16 20 24; do
  echo -n "[rccl queues $q] "; GPU_MAX_HW_QUEUES=$q PORESEG_BENCH_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + q + rep)) bench.py --gpus 1 --no-cpu --no-h2d --no-detail 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], end=" ")'
done; echo
done
