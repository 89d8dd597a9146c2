#!/usr/bin/env python3
"""numpy prototype of the GROUP bound of a window scan: an upper bound of the gain of every candidate
inside a group of GS samples from the exact sums at the group's two boundaries plus two numbers per group that K0 emits.

The gain of a candidate k of a window y[0..n) is a CONVEX function of the left part's (k, S1, S2):
    g = n log V_tot - k phi(S1/k, S2/k) - (n-k) phi((T1-S1)/(n-k), (T2-S2)/(n-k)),   phi(a, b) = log(b - a^2) concave,
and k phi(S1/k, S2/k) is the perspective of phi: jointly concave.  The prefix path (k, S1(k), S2(k)), P <= k <= Q, of a
group lies in the parallelepiped  chord(k) + (0, d1, d2z + 2 mu_g d1),  |d1| <= D1, |d2z| <= D2, with
    d1(j)  = sum_{i<j} (y_i - mu_g)                    (bridge of the centred samples; mu_g the group's mean)
    d2z(j) = sum_{i<j} ((y_i - mu_g)^2 - v_g)          (bridge of their squares; v_g the group's variance)
so the gain inside the group is at most the largest gain at the eight vertices: k in {P, Q}, d1 = +-D1, d2z = +-D2.  At a
vertex everything is known from the boundary evaluation; the displacement changes
    SS_L by  d2z + 2 (mu_g - mu_L) d1 - d1^2 / k,       SS_R by  -d2z - 2 (mu_g - mu_R) d1 - d1^2 / (n-k),
and with log(1 + t) >= t - c(T) t^2 on |t| <= T, c(T) = (-T - log(1 - T)) / T^2 <= 0.5 + 0.44 T for T <= 0.3 (c is convex; the
modes before "sym" keep the constant 0.537 and T <= 0.1), the cost falls by at most
    A = D2 |1/V_L - 1/V_R| + 2 D1 |(mu_g-mu_L)/V_L - (mu_g-mu_R)/V_R| + D1^2 (1/SS_L + 1/SS_R) + c(tL) k tL^2 + c(tR) (n-k) tR^2,
tL = (D2 + 2 |mu_g-mu_L| D1 + D1^2/k) / SS_L  (tR alike).  Group bound = max over the two boundaries of gain + A (log2 e).
The first-order term keeps the cancellation between the two sides (1/V_L - 1/V_R is ~ sqrt(2/k) / sigma^2 on noise).

K0 cannot afford per-sample bridges; it has S1, S2, min, max per 8-sample block.  D1, D2 from those, first form ("block",
"group", and "sym" = "group" with one slack per boundary):
    D1 <= max over block boundaries |d1| + 2 max_b (ymax_b - ymin_b)         (a centred partial sum of j of 8 values)
    D2 <= max over block boundaries |d2z| + 2 max_b max(|ymax_b - mu_g|, |ymin_b - mu_g|)^2
second form ("tight"; "kernel" = "tight" with one slack per boundary: what the kernels do), with q_b = sum over block b of
(y - mu_g)^2 = s2_b - mu_g (2 s1_b - 8 mu_g):
    D1 <= max_b (max(|d1(8b)|, |d1(8b+8)|) + sqrt(2 q_b))                   (Cauchy-Schwarz on the block-centred samples)
    D2 <= max_b (max(|d2z(8b)|, |d2z(8b+8)|) + 7/8 q_b)                     (the squares are non-negative)
The script runs the reference recursion on a synthetic trace, and for every window it scans: checks that the bound is
never below the largest interior gain, and counts how many 63-block rows of the fine sweep stay live.
usage: group_bound.py [n_samples] [seed]"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pypore_amd import synth          # noqa: E402

LOG2E = 1.4426950408889634
MW, W, MAXW = 100, 10000, 1000000
THR = 18.4204807339517


def window_gains(c1, c2, ps, pe):
    """log2-unit gains of candidates k = 1..n-1 of window [ps, pe) (c1, c2: exclusive prefix sums of the event)"""
    n = pe - ps
    k = np.arange(1, n)
    a1 = c1[ps + k] - c1[ps]; a2 = c2[ps + k] - c2[ps]
    T1 = c1[pe] - c1[ps]; T2 = c2[pe] - c2[ps]
    ssl = a2 - a1 * a1 / k
    ssr = (T2 - a2) - (T1 - a1) ** 2 / (n - k)
    sst = T2 - T1 * T1 / n
    with np.errstate(divide="ignore", invalid="ignore"):
        g = n * np.log2(sst / n) - k * np.log2(ssl / k) - (n - k) * np.log2(ssr / (n - k))
    return g, ssl, ssr, a1, a2, T1, T2


def rec_windows(y):
    """the reference recursion (cparsers.pyx:180-203) with memoised left children; yields every window it scans"""
    n = y.size
    c1 = np.concatenate(([0.0], np.cumsum(y))); c2 = np.concatenate(([0.0], np.cumsum(y * y)))
    wins = []
    bounds = []

    def best(ps, pe):
        if pe - ps <= 2 * MW:
            return -1
        g = window_gains(c1, c2, ps, pe)[0]
        gg = g[MW - 1:pe - ps - MW] / LOG2E
        i = int(np.argmax(gg)) if gg.size else -1
        hit = i >= 0 and gg[i] > THR
        wins.append((ps, pe, ps + MW + i if hit else -1))
        return ps + MW + i if hit else -1

    stack = [(0, n, 0)]
    while stack:
        start, end, j0 = stack.pop()
        split = -1
        ps = start + j0 * (W // 2)
        while ps < end - 2 * MW:
            split = best(ps, min(end, ps + W))
            if split >= 0:
                break
            ps += W // 2
        if split == -1:
            continue
        bounds.append(split)
        jl = max(0, (split - W - start) // (W // 2) + 1) if split - W - start >= 0 else 0
        stack.append((split, end, 0))
        stack.append((start, split, jl))
    return c1, c2, wins, sorted(bounds)


def group_data(y, c1, c2, P, GS, mode):
    """(D1, D2) of the group of samples [P, P+GS) (absolute positions) -- what K0 would emit"""
    seg = y[P:P + GS]
    mu = seg.mean()
    z = seg - mu
    vg = float(np.dot(z, z)) / GS
    d1 = np.cumsum(z)[:-1]
    d2 = np.cumsum(z * z - vg)[:-1]
    if mode == "sample":
        return np.abs(d1).max(), np.abs(d2).max()
    if mode == "tight":
        # what K0 does (PS_K0_AMP = 2): bridges at the block boundaries, inside block b at most sqrt(2 q_b) / 7/8 q_b off the
        # chord between its two boundaries, q_b = the block's sum of squares about the group's mean
        e1 = np.abs(np.concatenate(([0.0], np.cumsum(z))))[::8]
        e2 = np.abs(np.concatenate(([0.0], np.cumsum(z * z - vg))))[::8]
        qb = (z.reshape(-1, 8) ** 2).sum(axis=1)
        return (np.maximum(e1[:-1], e1[1:]) + np.sqrt(2.0 * qb)).max(), (np.maximum(e2[:-1], e2[1:]) + 0.875 * qb).max()
    # block mode: bridges at the block boundaries + slack from the blocks' min / max
    b = seg.reshape(-1, 8)
    d1b = np.abs(d1[7::8]).max() if GS > 8 else 0.0
    d2b = np.abs(d2[7::8]).max() if GS > 8 else 0.0
    M = max(seg.max() - mu, mu - seg.min())
    if mode == "block":
        R = (b.max(axis=1) - b.min(axis=1)).max()
    else:                                                  # "group": only the group's min / max
        R = seg.max() - seg.min()
    return d1b + 2.0 * R, d2b + 2.0 * M * M


def group_bound(n, g, ssl, ssr, a1, T1, k, other, D1, D2, GS):
    """gain + A at boundary k (window-relative, 1 <= k <= n-1) for the group between k and `other`"""
    i = k - 1
    VL = ssl[i] / k; VR = ssr[i] / (n - k)
    if not (VL > 0 and VR > 0):
        return np.inf
    muL = a1[i] / k; muR = (T1 - a1[i]) / (n - k)
    lo, hi = min(k, other), max(k, other)
    mug = (a1[hi - 1] - a1[lo - 1]) / GS
    first = D2 * abs(1 / VL - 1 / VR) + 2 * D1 * abs((mug - muL) / VL - (mug - muR) / VR) + D1 * D1 * (1 / ssl[i] + 1 / ssr[i])
    tL = (D2 + 2 * abs(mug - muL) * D1 + D1 * D1 / k) / ssl[i]
    tR = (D2 + 2 * abs(mug - muR) * D1 + D1 * D1 / (n - k)) / ssr[i]
    if tL > 0.1 or tR > 0.1:
        return np.inf
    second = 0.537 * (k * tL * tL + (n - k) * tR * tR)
    return g[i] + LOG2E * (first + second)


def boundary_A(n, ssl, ssr, a1, T1, k, groups):
    """one A for boundary k that serves both adjacent groups: `groups` = [(D1, D2, mu_g), ...] (one or two)"""
    i = k - 1
    VL = ssl[i] / k; VR = ssr[i] / (n - k)
    if not (VL > 0 and VR > 0):
        return np.inf
    muL = a1[i] / k; muR = (T1 - a1[i]) / (n - k)
    D1 = max(gr[0] for gr in groups); D2 = max(gr[1] for gr in groups)
    c1 = max(abs((gr[2] - muL) / VL - (gr[2] - muR) / VR) for gr in groups)
    wL = max(abs(gr[2] - muL) for gr in groups); wR = max(abs(gr[2] - muR) for gr in groups)
    first = D2 * abs(1 / VL - 1 / VR) + 2 * D1 * c1 + D1 * D1 * (1 / ssl[i] + 1 / ssr[i])
    tL = (D2 + 2 * wL * D1 + D1 * D1 / k) / ssl[i]
    tR = (D2 + 2 * wR * D1 + D1 * D1 / (n - k)) / ssr[i]
    if tL > 0.3 or tR > 0.3:
        return np.inf
    # (log(1 + t) >= t - c(T) t^2 on |t| <= T with c(T) = (-T - log(1 - T)) / T^2, convex: below 0.5 + 0.4325 T on [0, 0.3])
    return LOG2E * (first + (0.5 + 0.44 * tL) * k * tL * tL + (0.5 + 0.44 * tR) * (n - k) * tR * tR)


def run(y, c1, c2, wins, GS, mode):
    """the bound on every group of every window in `wins`; returns counts (violations, groups, live rows of the fine sweep)"""
    thr2 = THR * LOG2E
    viol = 0
    tot_groups = kept_groups = 0
    rows_now = rows_live = 0
    nosplit = split = 0
    lr_ns = lr_s = 0
    loose = []
    for (ps, pe, res) in wins:
        n = pe - ps
        if n < 4 * GS:
            continue
        g, ssl, ssr, a1, a2, T1, T2 = window_gains(c1, c2, ps, pe)
        dlt = 0.02 + 8.0e-6 * n
        dthr = dlt + 3.0e-6 * n + 1.0e-6 * thr2
        # groups aligned to the absolute sample index (global block index in the kernel)
        P0 = -(-ps // GS) * GS
        Ps = np.arange(P0, pe - GS + 1, GS)
        Ps = Ps[(Ps - ps >= 1) & (Ps + GS - ps <= n - 1)]
        if Ps.size == 0:
            continue
        kb = np.concatenate((Ps, [Ps[-1] + GS])) - ps          # coarse boundaries, window-relative
        inr = (kb >= MW) & (kb <= n - MW)
        bm = g[kb[inr] - 1].max() if inr.any() else -np.inf
        Tprune = max(thr2 - dthr, bm - 2 * dlt) - 2 * dlt
        nblk = n // 8
        rows = (nblk + 62) // 63
        live = np.zeros(rows, dtype=bool)
        # blocks outside the coarse coverage: always swept
        first_cov, last_cov = kb[0], kb[-1]
        live[0: (first_cov // 8) // 63 + 1] = True
        live[min(rows - 1, (last_cov // 8) // 63):] = True
        if mode in ("sym", "kernel"):
            gd = []
            for P in Ps:
                D1, D2 = group_data(y, c1, c2, P, GS, "tight" if mode == "kernel" else "group")
                gd.append((D1, D2, (a1[P - ps + GS - 1] - a1[P - ps - 1]) / GS))
            GA = []
            for ci, k in enumerate(kb):
                adj = [gd[j] for j in (ci - 1, ci) if 0 <= j < len(gd)]
                GA.append(g[k - 1] + boundary_A(n, ssl, ssr, a1, T1, k, adj))
        for gi, P in enumerate(Ps):
            kP, kQ = P - ps, P - ps + GS
            if mode in ("sym", "kernel"):
                hb = max(GA[gi], GA[gi + 1])
            else:
                D1, D2 = group_data(y, c1, c2, P, GS, mode)
                hb = max(group_bound(n, g, ssl, ssr, a1, T1, kP, kQ, D1, D2, GS),
                         group_bound(n, g, ssl, ssr, a1, T1, kQ, kP, D1, D2, GS))
            inner = g[kP:kQ - 1].max() if kQ - 1 > kP else -np.inf              # candidates kP+1 .. kQ-1, in range or not
            if inner > hb + 1e-9:
                viol += 1
            tot_groups += 1
            if res < 0 and np.isfinite(hb):
                loose.append(hb - max(g[kP - 1], g[kQ - 1]))
            if not hb < Tprune:
                kept_groups += 1
                live[min(rows - 1, (kP // 8) // 63): min(rows - 1, ((kQ + 7) // 8) // 63) + 1] = True
        rows_now += rows
        rows_live += int(live.sum())
        if res < 0:
            nosplit += 1; lr_ns += int(live.sum())
        else:
            split += 1; lr_s += int(live.sum())
    return dict(violations=viol, groups=tot_groups, kept=kept_groups, rows=rows_now, rows_live=rows_live, windows=nosplit + split,
                nosplit=nosplit, live_rows_nosplit=lr_ns / max(nosplit, 1), live_rows_split=lr_s / max(split, 1),
                loose=np.array(loose))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
    y = synth.random_dwell_counts(N, seed).astype(np.float64)
    y -= y[0]
    c1, c2, wins, bounds = rec_windows(y)
    print("trace %d samples, %d windows scanned, %d boundaries" % (N, len(wins), len(bounds)))
    for GS in (64, 128, 256, 512):
        for mode in ("sample", "block", "group", "sym", "kernel"):
            if mode in ("sym", "kernel") and GS == 64:
                continue
            r = run(y, c1, c2, wins, GS, mode)
            lo = r["loose"]
            print("GS %3d %-6s: violations %d | groups kept %.2f %% | fine rows live %.1f %% (%d of %d; %d windows: %d no split)"
                  " | bound - max boundary gain on no-split windows: median %.2f  90%% %.2f  99%% %.2f" %
                  (GS, mode, r["violations"], 100.0 * r["kept"] / max(r["groups"], 1), 100.0 * r["rows_live"] / max(r["rows"], 1),
                   r["rows_live"], r["rows"], r["windows"], r["nosplit"], np.median(lo), np.quantile(lo, 0.9), np.quantile(lo, 0.99)))
            print("      live rows per window: no split %.2f, split %.2f" % (r["live_rows_nosplit"], r["live_rows_split"]))


if __name__ == "__main__":
    main()
