"""Randomised parity fuzz (GPU box): HIP path vs the CPU oracle on many seeded traces and parameter sets, in the
default and verify modes, fp32 and int16 input, events at odd offsets.  Not part of the pytest suite (minutes).
FUZZ_SCALE=64: the same signals on a grid 64 times finer (fp32 only): counts run to 2^21 from the event's first sample,
beyond the 32-bit digest -- the calls take the 64-bit digest (or, with PORESEG_WIDE_BS=0, the LDS-window kernels)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none

ctx = engine.context(0)
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
t0 = time.time()
bad = 0
SCALE = int(os.environ.get("FUZZ_SCALE", "1"))
Q = synth.QUANTUM / SCALE
routes = {}
shifted = unchecked = cross = same_route = verify_outside = 0
def near_threshold_outside_domain(msg, evs, params):
    """a verify-mode disagreement is downgraded to a note only when it can be the reference's own rounding --
    the window the message names, taken from an event OUTSIDE the reference's exact domain, has an exact best gain (oracle
    on the shifted event, whose sums are exact) within 1e-4 relative of the threshold, or its two best gains that close to
    each other.  Anything else stays an error: a pruning mistake on the 64-bit digest must not hide behind the domain."""
    import re
    m = re.search(r"window \[(\d+),(\d+)\)", msg)
    if not m:
        return False
    ps, pe = int(m.group(1)), int(m.group(2))
    thr = oracle.min_gain(**{k_: v_ for k_, v_ in params.items()})
    for k in evs:
        if len(k) * float(np.abs(k).max()) ** 2 < 2.0 ** 53 or pe > len(k):
            continue
        if (pe - ps) * float(np.abs(k[ps:pe] - k[ps]).max()) ** 2 >= 2.0 ** 53:
            return True                                  # not even the window's own sums are exact: nothing to compare with
        _, sc = oracle.score_window((k[ps:pe] - k[ps]).astype(np.float64) * Q, params["min_width"], 0.0)
        top = np.sort(sc[np.isfinite(sc)])[-2:] if np.isfinite(sc).any() else np.zeros(0)
        if top.size and (abs(top[-1] - thr) <= 1e-4 * max(1.0, abs(thr)) or (top.size == 2 and top[-1] - top[-2] <= 1e-4 * max(1.0, abs(top[-1])))):
            return True
    return False


for seed in range(n_seeds):
    rng = np.random.RandomState(10_000 + seed + int(os.environ.get("FUZZ_BASE", "0")))
    mw = int(rng.choice([8, 20, 100, 250]))
    W = int(max(2 * mw, rng.choice([400, 1000, 4000, 10000, 25000])))
    maxw = int(rng.choice([W, 3 * W, 50000, 1000000]))
    params = dict(min_width=mw, max_width=max(maxw, mw), window_width=W,
                  prior_segments_per_second=float(rng.choice([1., 10., 100.])))
    n_ev = int(rng.choice([1, 1, 3, 7]))
    sigma = float(rng.choice([0.0, 1.0, 4.0, 30.0, 150.0]))
    dc = int(rng.choice([0, 0, 500, -3000, 9000]))
    evs = []
    for e in range(n_ev):
        n = int(rng.randint(3000, 300000 if n_ev == 1 else 60000))
        lo = int(rng.randint(50, 3000)); hi = lo + int(rng.randint(100, 30000))
        k = np.empty(n, dtype=np.int64); i = 0
        while i < n:
            d = int(rng.randint(lo, hi)); lvl = int(rng.randint(-2500, 2500))
            k[i:i + d] = lvl; i += d
        if sigma > 0:
            k += np.rint(rng.normal(0.0, sigma, n)).astype(np.int64)
        k = np.clip(k + dc, -32000, 32000) * SCALE
        if SCALE > 1:
            k = k + rng.randint(-(SCALE // 2), SCALE // 2 + 1, n)      # use the fine grid's low bits too
        evs.append(k)
    use_i16 = bool(rng.randint(0, 2)) and SCALE == 1
    pad = [int(rng.randint(0, 9)) for _ in range(n_ev)]          # odd offsets between events
    total = sum(len(k) + p for k, p in zip(evs, pad)) + 16
    buf = np.zeros(total, dtype=np.int64); starts = []; lens = []; pos = 0
    for k, p in zip(evs, pad):
        pos += p; starts.append(pos); lens.append(len(k)); buf[pos:pos + len(k)] = k; pos += len(k)
    if use_i16:
        dev = torch.from_numpy(buf.astype(np.int16)).cuda()
    else:
        dev = torch.from_numpy((buf.astype(np.float64) * Q).astype(np.float32)).cuda()
    sp = _lib.split_params(**params)
    # The reference's own prefix sums are exact only while n_event * max|k|^2 < 2^53.  Beyond that
    # (possible with FUZZ_SCALE > 1) its rounding decides near-ties, so the check is against the oracle on the same
    # event with its first sample subtracted -- the gains are shift invariant and those sums are exact -- if that is
    # inside the domain; an event outside even that (none of the reference's sums is exact: its decisions on near ties are
    # its own rounding noise) is compared between the two device routes that share no scan code -- the block-sum scan on
    # the 64-bit digest and the LDS-window kernels (wide_bs = 0) -- which both form exact window-local sums.
    refs = []
    for k in evs:
        if len(k) * float(np.abs(k).max()) ** 2 < 2.0 ** 53:
            refs.append(oracle.parse(k.astype(np.float64) * Q, **params))
        elif len(k) * float(np.abs(k - k[0]).max()) ** 2 < 2.0 ** 53:
            refs.append(oracle.parse((k - k[0]).astype(np.float64) * Q, **params)); shifted += 1
        else:
            refs.append(None); unchecked += 1
    for mode in (0, 2):
        ctx.set_option("mode", mode)
        try:
            b, boff, _ = ctx.segment_events(dev, np.array(starts, dtype=np.int64), np.array(lens, dtype=np.int64), sp,
                                            Q, want_stats=False)
            routes[ctx.timings()["wide_redo"]] = routes.get(ctx.timings()["wide_redo"], 0) + 1
            b = b.cpu().numpy()
            for e in range(n_ev):
                got = b[boff[e]:boff[e + 1]]
                if refs[e] is not None and not np.array_equal(got, refs[e]):
                    bad += 1
                    print("MISMATCH seed", seed, "mode", mode, "event", e, params, "sigma", sigma, "dc", dc, "i16", use_i16,
                          "got", len(got), "ref", len(refs[e]))
        except Exception as ex:
            # Verify mode compares the screen's decision -- taken from exact sums about the event's first sample -- with a
            # whole-window scan in the REFERENCE's arithmetic (raw fp64 sums).  For an event outside the domain where that
            # arithmetic is exact (large DC level on a fine grid) the reference's own cancellation noise, ~1e-5 relative on
            # a gain, decides a window whose best gain lies that close to the threshold; the two then disagree although the
            # default-mode result equals the oracle on the shifted event (checked above).  Noted, not a problem; round 4's
            # validation found one such window in 5 000 seeds (gain 13.813478 exact / 13.813740 in raw fp64 / threshold 13.813510).
            if mode == 2 and "verify mode" in repr(ex) and near_threshold_outside_domain(repr(ex), evs, params):
                verify_outside += 1
                print("NOTE seed", seed, "verify-mode disagreement on a call with an event outside the reference's exact sums:", repr(ex)[60:230])
            else:
                bad += 1
                print("ERROR seed", seed, "mode", mode, params, "sigma", sigma, "dc", dc, "i16", use_i16, repr(ex)[:300])
    ctx.set_option("mode", 0)
    if any(r is None for r in refs):
        try:
            b1, o1, _ = ctx.segment_events(dev, np.array(starts, dtype=np.int64), np.array(lens, dtype=np.int64), sp, Q, want_stats=False)
            r1 = ctx.timings()["wide_redo"]
            ctx.set_option("wide_bs", 0)
            b2, o2, _ = ctx.segment_events(dev, np.array(starts, dtype=np.int64), np.array(lens, dtype=np.int64), sp, Q, want_stats=False)
            r2 = ctx.timings()["wide_redo"]
            b1, b2 = b1.cpu().numpy(), b2.cpu().numpy()
            for e in range(n_ev):
                if refs[e] is None:
                    # (a call whose windows never exceed 2 min_width scans nothing: K0 is not refused and both calls report
                    #  route 0 -- compared all the same, counted apart)
                    cross += r1 != r2
                    same_route += r1 == r2
                    if not np.array_equal(b1[o1[e]:o1[e + 1]], b2[o2[e]:o2[e + 1]]):
                        bad += 1
                        print("ROUTE MISMATCH seed", seed, "event", e, params, "routes", r1, r2)
        except Exception as ex:
            bad += 1
            print("ERROR (route check) seed", seed, repr(ex)[:300])
        finally:
            ctx.set_option("wide_bs", 1)
print("fuzz: %d seeds, %d problems, %.0f s, scale %d, calls per route (0 32-bit digest, 1 64-bit digest, 2 LDS-window) %s, "
      "events beyond the reference's exact sums: %d checked on the shifted event, %d beyond that too, of which %d compared "
      "between the 64-bit digest and the LDS-window route (%d more on one route both times: nothing to scan); verify-mode disagreements on such calls (the reference's "
      "own rounding decides there): %d; counters %s"
      % (n_seeds, bad, time.time() - t0, SCALE, routes, shifted, unchecked, cross, same_route, verify_outside, ctx.timings()))
sys.exit(1 if bad else 0)
