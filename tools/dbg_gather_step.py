"""Per-step host time of segmentation vs. boundary gather (one rank, RCCL initialised)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29513")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from pypore_amd import _lib, engine, synth, dist as pdist
import bench
ctx = engine.context(0)
n = 10**8
d = synth.dwell_table(2024, n); ends = np.cumsum(d)
lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
trace = ctx.synth_trace(n, 2024, ends, lv, dtype=torch.float32)
ev_off = np.array([0, n], dtype=np.int64)
params = _lib.split_params(**bench.PARAMS)
bg = pdist.BoundaryGather(32768, trace.device, torch.int32)
def run(mode, K=50):
    ts = np.zeros(3)
    pend = []
    for it in range(K + 5):
        t0 = time.perf_counter()
        b, o, st = ctx.segment_batch(trace, ev_off, params, synth.QUANTUM)
        t1 = time.perf_counter()
        if mode >= 1:
            pend.append(bg.submit(b))
        t2 = time.perf_counter()
        if mode >= 2:
            while len(pend) > 1:
                bg.result(pend.pop(0))
        t3 = time.perf_counter()
        if it >= 5:
            ts += (t1 - t0, t2 - t1, t3 - t2)
        if mode == 1:
            s = bg.slots[pend.pop(0) % 2]; s["work"] = None
    while pend:
        bg.result(pend.pop(0))
    print("mode %d: segment %.1f us, submit %.1f us, result(prev) %.1f us" % ((mode,) + tuple(ts / K * 1e6)))
run(0); run(1); run(2); run(0)
dist.destroy_process_group()
