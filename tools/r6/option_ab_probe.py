#!/usr/bin/env python3
"""Round 6: A/B of context options inside ONE process, interleaved -- the pool's step over K steps and over the driver's 20, and the
lone call's device time.  usage: option_ab_probe.py "name=v[,name=v..]" "name=v[,...]" ... [--T 16] [--K 100] [--pairs 8]
Each positional argument is one setting (options applied to every context of the pool); 'base' = nothing changed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import numpy as np, torch
from pypore_amd import _lib, engine, synth
engine.apply_env_defaults()
args = [a for a in sys.argv[1:] if a == "base" or "=" in a]
def flag(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default
T, K, PAIRS = flag("--T", 16), flag("--K", 100), flag("--pairs", 8)
settings = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",")) if a != "base" else {} for a in args]
names = sorted({k for s_ in settings for k in s_})
n = 100_000_000
params = _lib.split_params(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
pool = engine.StreamPool(0, T)
ctx0 = pool.contexts[0]
defaults = {"gather_fused": 1, "download_by_kernel": 1, "k0_unaligned": 1, "k0_sets": 2, "k0_admit": 3, "k0_waves": 1, "lat_help": 0, "k0_chain": 1}
traces = []
for t in range(T):
    sd = 2024 + 1000 * t
    d = synth.dwell_table(sd, n); ends = np.cumsum(d); lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
    traces.append(ctx0.synth_trace(n, sd, ends, lv, dtype=torch.float32))
ev_off = np.array([0, n], dtype=np.int64)
outs = [torch.empty(n // 100 + 1, dtype=torch.int32, device="cuda") for _ in range(T)]
job = lambda cx, k, t: cx.segment_batch(traces[t], ev_off, params, synth.QUANTUM, want_stats=False, out=outs[t])[0].numel()
import gc; gc.collect(); gc.freeze()
ref = pool.run(4 * T, job)[-T:]


POOL_OPTS = ("k0_sets", "k0_admit", "k0_waves", "lat_help", "k0_shared", "k0_chain")


def apply(setting):
    full = {k: setting.get(k, defaults.get(k, 0)) for k in names}
    # options of a shared chip go through the pool (it re-configures its contexts around every run); the others are set directly
    pool.overrides = {k: v for k, v in full.items() if k in POOL_OPTS}
    pool._shared_now.clear()
    for cx in pool.contexts:
        for k, v in full.items():
            if k not in POOL_OPTS:
                cx.set_option(k, v)


def measure(setting, steps):
    apply(setting)
    pool.run(T, job)
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = pool.run(steps, job); torch.cuda.synchronize()
    assert all(r[k] == ref[k % T] for k in range(len(r))), "boundary counts changed"
    return (time.perf_counter() - t0) / steps * 1e3


def lone(setting):
    # (right after measure(): the pool has given contexts[0] back as a lone context; the direct options are still set)
    for _ in range(3):
        job(ctx0, 0, 0)
    s = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        job(ctx0, 0, 0); s += ctx0.seq_ms()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 30 * 1e3, s / 30


res = {i: dict(k=[], k20=[], lone=[], dev=[]) for i in range(len(settings))}
for p in range(PAIRS):
    order = list(range(len(settings)))
    if p % 2:
        order.reverse()
    for i in order:
        res[i]["k"].append(measure(settings[i], K))
        res[i]["k20"].append(measure(settings[i], 20))
        a, b = lone(settings[i])
        res[i]["lone"].append(a); res[i]["dev"].append(b)
for i, a in enumerate(args):
    r = res[i]
    print("%-44s %d steps: mean %.4f median %.4f sd %.4f | 20 steps: mean %.4f median %.4f sd %.4f | lone call %.4f ms wall, %.4f device"
          % (a, K, np.mean(r["k"]), np.median(r["k"]), np.std(r["k"]), np.mean(r["k20"]), np.median(r["k20"]), np.std(r["k20"]),
             np.mean(r["lone"]), np.mean(r["dev"])))
