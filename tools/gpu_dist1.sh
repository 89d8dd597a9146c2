#!/bin/bash
# N > 1 code path of bench.py on the one GPU of the box: one rank, RCCL initialised, the boundary
# gather and the barriers in the timed region (under torchrun).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1
python bench.py > gpurun_out/dist1_plain.json 2> gpurun_out/dist1_plain.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
  --master-port 29510 bench.py --gpus 1 > gpurun_out/dist1_torchrun_nodist.json 2> gpurun_out/dist1_torchrun_nodist.err
PORESEG_BENCH_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
  --master-port 29511 bench.py --gpus 1 > gpurun_out/dist1_rccl.json 2> gpurun_out/dist1_rccl.err
python - <<'P'
import json
for f in ("plain","torchrun_nodist","rccl"):
    try:
        d=json.loads(open("gpurun_out/dist1_%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d["config"]["boundaries"])
    except Exception as e:
        print(f, "failed", e); print(open("gpurun_out/dist1_%s.err"%f).read()[-2000:])
P
