"""Randomised parity on sparse shapes (long stretches without splits: where the look-ahead kernel's helpers and the
forced-split arithmetic work): random lengths, dwell regimes, window widths, max_width (multiples of W/2 and not), min_width,
one to three events per call, dtype, bridge budget; every call against the oracle (one core).  usage: fuzz_sparse.py [seeds] [base]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
ctx = engine.context(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
base = int(sys.argv[2]) if len(sys.argv) > 2 else 0
problems = 0
t0 = time.time()
helped = 0
for seed in range(base, base + N):
    r = np.random.default_rng(seed)
    n_ev = int(r.integers(1, 4))
    W = int(r.choice([2000, 4000, 10000, 20000]))
    half = W // 2
    mw = int(r.choice([8, 50, 100, 400]))
    if 2 * mw >= W: mw = 50
    kind = int(r.integers(0, 5))
    maxw = [1000000, int(r.integers(20, 300)) * half, int(r.integers(10 * W, 400 * W)) | 1, 10 ** 9, int(r.integers(4, 40)) * W][kind]
    maxw = max(maxw, W + 2 * mw, mw + 1)
    kw = dict(min_width=mw, max_width=maxw, window_width=W, prior_segments_per_second=float(r.choice([10., 1., 100.])), sampling_freq=1e5)
    lens = [int(r.integers(200_000, 2_500_000)) for _ in range(n_ev)]
    regime = int(r.integers(0, 4))
    pieces = []
    for e, ln in enumerate(lens):
        lo, hi = [(ln + 1, ln + 2), (100000, 1000000), (300000, 3000000), (20000, 200000)][regime]
        d = synth.dwell_table(seed * 7 + e, ln, lo, hi)
        ends = np.cumsum(d); lv = synth.LEVEL_COUNTS[(np.arange(len(d)) + e) % 5].astype(np.int32)
        pieces.append(ctx.synth_trace(ln, seed * 7 + e, ends, lv, dtype=torch.float32))
    gaps = [int(r.integers(0, 3)) * 8 + int(r.integers(0, 8)) * (seed % 2) for _ in range(n_ev)]     # events at odd offsets every other seed
    total = sum(lens) + sum(gaps)
    dt = torch.int16 if r.integers(0, 3) == 0 else torch.float32
    buf = torch.zeros(total, dtype=torch.float32, device="cuda")
    starts, pos = [], 0
    for g_, p_, ln in zip(gaps, pieces, lens):
        pos += g_; buf[pos:pos + ln] = p_; starts.append(pos); pos += ln
    if dt == torch.int16:
        buf = torch.round(buf / synth.QUANTUM).to(torch.int16)
    ev_start = np.array(starts, dtype=np.int64)
    # ps_segment_batch takes contiguous events: pack them (ev_off), keeping each event's own samples
    packed = torch.cat([buf[s:s + ln] for s, ln in zip(starts, lens)])
    ev_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    budget = int(r.choice([256, 256, 3, 1]))
    ctx.set_option("bridge_budget", budget)
    ctx.set_option("lat_help", int(r.integers(0, 4) != 0))
    try:
        b, o, _ = ctx.segment_batch(packed, ev_off, _lib.split_params(**kw), synth.QUANTUM, want_stats=False)
    except Exception as ex:
        print("seed %d: ERROR %r %s" % (seed, ex, kw)); problems += 1; continue
    b = b.cpu().numpy()
    x = packed.cpu().numpy().astype(np.float64) * (synth.QUANTUM if dt == torch.int16 else 1.0)
    for e in range(n_ev):
        ref = oracle.parse(x[ev_off[e]:ev_off[e + 1]], **kw)
        got = b[o[e]:o[e + 1]]
        if not np.array_equal(got, ref):
            problems += 1
            print("seed %d event %d: DIFFERENT (%d vs %d boundaries) %s lens %s regime %d dtype %s budget %d" % (seed, e, len(got), len(ref), kw, lens, regime, dt, budget))
            break
ctx.set_option("bridge_budget", 256); ctx.set_option("lat_help", 1)
print("sparse fuzz: %d seeds from %d, %d problems, %.0f s" % (N, base, problems, time.time() - t0))
