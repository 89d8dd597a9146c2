#!/usr/bin/env python3
"""Round 6: ps_detect_segment_trace (one pass over a file trace) against the two calls it replaces -- ps_detect_events +
ps_segment_events, themselves fuzzed against the oracle by tools/fuzz_gpu.py and tests -- and, for the events, against the
oracle's restatement of lambda_event_parser: random trace lengths (any remainder mod 8, tiny ones), open-channel / blockade
levels, noise from none to wide enough that the current hovers about the threshold (many mixed blocks, thousands of edges:
the edge kernel's slow path and the overflow of its edge list), gaps and event lengths, thresholds, min_duration, ADC offsets,
int16 / float32, statistics on and off.  usage: fuzz_single_pass.py [seeds] [base]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from pypore_amd import _lib, engine, synth
engine.apply_env_defaults()
ctx = engine.context(0)
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
base = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = fell_back = n_events = 0
stage = ""
t0 = time.time()
for seed in range(n_seeds):
    rng = np.random.RandomState(99_000 + base + seed)
    n = int(rng.choice([rng.randint(1, 64), rng.randint(64, 5000), rng.randint(5000, 300000), rng.randint(300000, 3000000)]))
    open_k = int(rng.randint(2500, 4000))
    thr_k = open_k - int(rng.randint(200, 900))                     # threshold in counts (a multiple of the quantum, or half a count off)
    thr = (thr_k + float(rng.choice([0.0, 0.0, 0.5, -0.25]))) * synth.QUANTUM
    sigma = float(rng.choice([0.0, 3.0, 32.0, 32.0, 150.0, 400.0]))
    c = np.empty(n, dtype=np.int64)
    i = 0
    inside = bool(rng.rand() < 0.3)                                 # a trace may begin inside a blockade
    while i < n:
        if inside:
            ln = int(rng.choice([rng.randint(1, 40), rng.randint(40, 3000), rng.randint(3000, 400000)]))
            j = i
            while j < min(n, i + ln):
                d = int(rng.randint(30, 20000))
                c[j:j + d] = int(rng.randint(200, max(201, thr_k - int(2 * sigma) - 50))) if rng.rand() < 0.9 else thr_k - int(rng.randint(-20, 60))
                j += d
        else:
            ln = int(rng.choice([rng.randint(1, 40), rng.randint(40, 3000), rng.randint(3000, 100000)]))
            c[i:i + ln] = open_k if rng.rand() < 0.9 else thr_k + int(rng.randint(-10, 40))     # (now and then: hovering about the threshold)
        i += ln
        inside = not inside
    if sigma > 0:
        c += np.round(rng.normal(0, sigma, n)).astype(np.int64)
    oc = int(rng.choice([0, 0, 1234, -700]))
    dtype = "int16" if rng.rand() < 0.6 else "float32"
    if dtype == "int16":
        raw = np.clip(c - oc, -32000, 32000)                        # (ps_sample_format: count = raw value + offset_counts)
        c = raw + oc
        t = torch.from_numpy(raw.astype(np.int16)).cuda()
    else:
        oc = 0
        t = torch.from_numpy((c * synth.QUANTUM).astype(np.float32)).cuda()
    mw = int(rng.choice([8, 20, 100, 250]))
    W = int(max(2 * mw, rng.choice([400, 1000, 4000, 10000])))
    params = _lib.split_params(min_width=mw, max_width=int(rng.choice([3 * W, 50000, 1000000])), window_width=W,
                               prior_segments_per_second=float(rng.choice([1., 10., 100.])))
    kw = dict(threshold=thr, min_duration=int(rng.choice([0, 10, 1000, 100000])), min_current=float(rng.choice([-0.5, -1e9, 20.0])),
              offset_counts=oc)
    want_stats = bool(rng.rand() < 0.4)
    try:
        st, ln_, b, off, stats = ctx.detect_segment_trace(t, synth.QUANTUM, params, want_stats=want_stats, **kw)
        fell_back += int(ctx.timings()["wide_redo"] == 3)
        st2, ln2 = ctx.detect_events(t, synth.QUANTUM, kw["threshold"], kw["min_duration"], kw["min_current"], oc)
        b2, off2, stats2 = ctx.segment_events(t, st2, ln2, params, synth.QUANTUM, oc, want_stats)
        stage = "events" if not (np.array_equal(st, st2) and np.array_equal(ln_, ln2)) else \
            "bounds" if not (np.array_equal(off, off2) and np.array_equal(b.cpu().numpy(), b2.cpu().numpy())) else ""
        ok = stage == ""
        if not ok and stage == "events":
            print("   fused", list(zip(st[:6].tolist(), ln_[:6].tolist())), len(st), "two calls", list(zip(st2[:6].tolist(), ln2[:6].tolist())), len(st2))
        if not ok and stage == "bounds":
            bb, bb2 = b.cpu().numpy(), b2.cpu().numpy()
            for e in range(len(st)):
                if not np.array_equal(bb[off[e]:off[e + 1]], bb2[off2[e]:off2[e + 1]]):
                    print("   event", e, (int(st[e]), int(ln_[e])), "fused", bb[off[e]:off[e + 1]][:8], "two calls", bb2[off2[e]:off2[e + 1]][:8])
                    break
        if ok and want_stats and len(st):
            s1, s2 = stats.cpu().numpy(), stats2.cpu().numpy()
            ok = np.allclose(s1[:, 0], s2[:, 0], rtol=1e-11, atol=1e-9) and np.allclose(s1[:, 1], s2[:, 1], rtol=1e-6, atol=1e-7) and \
                np.array_equal(s1[:, 2:], s2[:, 2:])
            if not ok:
                stage = "stats"
                k = int(np.argmax(~(np.isclose(s1, s2, rtol=1e-6, atol=1e-7).all(axis=1))))
                print("   row", k, s1[k], s2[k])
        if ok and n <= 400000:
            x = c.astype(np.float64) * synth.QUANTUM
            rs, rl = oracle.lambda_events(x, threshold=thr, min_duration=kw["min_duration"], min_current=kw["min_current"])
            ok = np.array_equal(st, rs) and np.array_equal(ln_, rl)
            if not ok:
                stage = "oracle events"
                print("   gpu", list(zip(st[:6].tolist(), ln_[:6].tolist())), len(st), "oracle", list(zip(rs[:6].tolist(), rl[:6].tolist())), len(rs))
        n_events += len(st)
    except Exception as e:                                          # noqa: BLE001 -- a fuzz reports what it meets
        ok = False
        print("seed", seed, "raised", repr(e)[:300])
    if not ok:
        bad += 1
        print("MISMATCH seed", base + seed, stage, dict(n=n, dtype=dtype, sigma=sigma, thr=thr, mw=mw, W=W, **kw), flush=True)
print("single-pass fuzz: %d seeds from %d, %d problems, %d events, %d calls took the two calls by themselves (wide range), %.0f s"
      % (n_seeds, base, bad, n_events, fell_back, time.time() - t0))
