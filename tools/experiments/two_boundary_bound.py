#!/usr/bin/env python3
"""numpy prototype of the two-boundary bound for the 8-sample blocks of a window scan (not in the kernels).

For a window y[0..n) and a candidate k (left part y[:k]) the screened gain in log2 units is
    g(k) = n log2(SS_tot / n) - k log2(SS_L(k) / k) - (n - k) log2(SS_R(k) / (n - k)),
SS = sum of squared deviations from the part's own mean.  The sweep evaluates g at the block boundaries k = 8 t and needs
an upper bound of g over the 7 candidates inside a block (p, q), q = p + 8.

  corner bound (seg_bs.hpp today): SS_L(k) >= SS_L(p), SS_R(k) >= SS_R(q)  -- both at their minimum, which no k realises.
  two-boundary bound (here): SS_L(k) + SS_R(k) = SS_tot - B(k), B(k) = N(k)^2 / (n k (n - k)), N(k) = n S1_L(k) - k T1, and
      |N(k)| <= |N(p)| + n sqrt(7 Q),  Q = sum over the block of (y - T1/n)^2            (Cauchy-Schwarz)
  so (SS_L, SS_R) lies in the box [SS_L(p), SS_L(q)] x [SS_R(q), SS_R(p)] on or above the line SS_L + SS_R = SS_tot - Bmax.
  The cost k log2(SS_L/k) + (n-k) log2(SS_R/(n-k)) increases in both arguments and is concave along the line and in k:
  its minimum is at an end of the line segment and at k = p + 1 or q - 1.

The script checks validity (bound >= the largest interior gain, every block, many windows of several kinds) and prints how
tight the two bounds are and how many blocks a window without a split would keep at the bench threshold.
usage: two_boundary_bound.py [trials]"""
import sys
import numpy as np

LOG2E = 1.4426950408889634


def prefix(y):
    c1 = np.concatenate(([0.0], np.cumsum(y)))
    c2 = np.concatenate(([0.0], np.cumsum(y * y)))
    return c1, c2


def gains(y):
    n = y.size
    c1, c2 = prefix(y)
    T1, T2 = c1[-1], c2[-1]
    k = np.arange(1, n)
    ssl = c2[k] - c1[k] ** 2 / k
    ssr = (T2 - c2[k]) - (T1 - c1[k]) ** 2 / (n - k)
    sst = T2 - T1 * T1 / n
    with np.errstate(divide="ignore", invalid="ignore"):
        g = n * np.log2(sst / n) - k * np.log2(ssl / k) - (n - k) * np.log2(ssr / (n - k))
    return k, g, ssl, ssr, sst, c1, c2


def bounds(y):
    """per block (p, q = p + 8), p = 8, 16, ...: largest interior gain, corner bound, two-boundary bound"""
    n = y.size
    k, g, ssl, ssr, sst, c1, c2 = gains(y)
    G = np.full(n + 1, -np.inf); G[1:n] = g
    SSL = np.zeros(n + 1); SSL[1:n] = ssl
    SSR = np.zeros(n + 1); SSR[1:n] = ssr
    T1 = c1[-1]; mu = T1 / n
    out = []
    for p in range(8, n - 16, 8):
        q = p + 8
        inner = G[p + 1:q].max()
        # corner: u = SS_L(p), v = SS_R(q), worst k in {p+1, q-1} (concave in k)
        def cost(kk, u, v):
            return kk * np.log2(u / kk) + (n - kk) * np.log2(v / (n - kk))
        if SSL[p] <= 0 or SSR[q] <= 0:
            continue
        c0 = n * np.log2(sst / n)
        corner = c0 - min(cost(p + 1, SSL[p], SSR[q]), cost(q - 1, SSL[p], SSR[q]))
        # two boundaries
        Np = n * c1[p] - p * T1
        Nq = n * c1[q] - q * T1
        blk = y[p:q] - mu
        Q = float(np.dot(blk, blk))
        Nmax = min(abs(Np), abs(Nq)) + n * np.sqrt(7.0 * Q)
        Kmin = min(p * (n - p), q * (n - q))
        Bmax = Nmax * Nmax / (n * Kmin)
        C = sst - Bmax                                     # SS_L + SS_R >= C inside the block
        u0, u1, v0, v1 = SSL[p], SSL[q], SSR[q], SSR[p]
        if u0 + v0 >= C:
            pts = [(u0, v0)]
        else:
            pts = [(max(u0, C - v1), min(v1, C - u0)), (min(u1, C - v0), max(v0, C - u1))]
        two = c0 - min(cost(kk, u, v) for (u, v) in pts for kk in (p + 1, q - 1))
        # the same from what a lane of the sweep has in registers (no block sums: Q from the growth of SS_L over the block)
        Bp, Bq = sst - SSL[p] - SSR[p], sst - SSL[q] - SSR[q]
        DL = SSL[q] - SSL[p]
        Qb = 2.0 * (1.0 + 8.0 / p) * DL + 16.0 * Bp * (n - p) / (n * p)
        kap = 1.0 + 8.0 * max(1.0 / p, 1.0 / (n - q))
        Bm = (np.sqrt(max(min(Bp, Bq), 0.0) * kap) + np.sqrt(max(7.0 * n * Qb / Kmin, 0.0))) ** 2
        s1, s2 = max(Bm - Bp, 0.0), max(Bm - Bq, 0.0)
        vv = max(SSR[p] - s1, SSR[q]); uu = max(SSL[q] - s2, SSL[p])
        e1 = LOG2E * (SSR[p] - vv) / vv; e2 = LOG2E * (SSL[q] - uu) / uu
        lgL = lambda kk: np.log2(SSL[kk] / kk); lgR = lambda kk: np.log2(SSR[kk] / (n - kk))
        Pp = G[p] + 7.0 * abs(lgL(p) - lgR(p)) + 49.0 * LOG2E * (1.0 / p + 1.0 / (n - p))
        Pq = G[q] + 7.0 * abs(lgL(q) - lgR(q)) + 49.0 * LOG2E * (1.0 / q + 1.0 / (n - q))
        lane = max(Pp + (n - p) * e1, Pq + q * e2)
        # ... and in the kernel's arithmetic: D = n S2 - S1^2 exact, everything after its conversion in float32, with the
        # margins of bs_block_bound2 (seg_bs.hpp)
        f = np.float32
        def ev(kk):
            DLk = kk * c2[kk] - c1[kk] ** 2; DRk = (n - kk) * (c2[-1] - c2[kk]) - (T1 - c1[kk]) ** 2
            r = (f(1) / f(kk), f(1) / f(n - kk))
            u = (f(DLk) * r[0] * r[0], f(DRk) * r[1] * r[1])
            c0f = np.log2(f(n * c2[-1] - T1 * T1) * (f(1) / f(n)) * (f(1) / f(n)))
            lg = (f(np.log2(u[0])) - f(c0f), f(np.log2(u[1])) - f(c0f))
            return dict(u=u, r=r, lg=lg, g=f(-(f(kk) * lg[0] + f(n - kk) * lg[1])))
        ep, eq = ev(p), ev(q)
        rn = f(1) / f(n); SStf = f(n * c2[-1] - T1 * T1) * rn; nf = f(n)
        pf, qf, npf, nqf = f(p), f(q), f(n - p), f(n - q)
        SSLp, SSRp, SSLq, SSRq = pf * ep["u"][0], npf * ep["u"][1], qf * eq["u"][0], nqf * eq["u"][1]
        eB = SStf * f(2.0e-6)
        Bpf, Bqf = (SStf - SSLp) - SSRp, (SStf - SSLq) - SSRq
        Bpu, Bqu = max(Bpf, f(0)) + eB, max(Bqf, f(0)) + eB
        DLb = max(SSLq - SSLp, f(0)) + eB
        Qf = f(2) * (f(8) * ep["r"][0] + f(1)) * DLb + f(16) * Bpu * (npf * rn) * ep["r"][0]
        rK = f(1) / min(pf * npf, qf * nqf)
        kapf = f(8) * max(ep["r"][0], eq["r"][1]) + f(1)
        sBQ = np.sqrt(min(Bpu, Bqu) * kapf, dtype=f) + np.sqrt(f(7) * nf * Qf * rK, dtype=f)
        Bmf = sBQ * (sBQ * f(1.00001)) + eB
        s1f, s2f = max(Bmf - (Bpf - eB), f(0)), max(Bmf - (Bqf - eB), f(0))
        v1, u2 = max(SSRp - s1f, SSRq), max(SSLq - s2f, SSLp)
        L2 = f(LOG2E) * f(1.00001)
        e1f, e2f = L2 * (SSRp - v1) * (f(1) / v1), L2 * (SSLq - u2) * (f(1) / u2)
        Ppf = ep["g"] + (f(7) * abs(ep["lg"][0] - ep["lg"][1]) + f(49.0 * LOG2E) * (ep["r"][0] + ep["r"][1]))
        Pqf = eq["g"] + (f(7) * abs(eq["lg"][0] - eq["lg"][1]) + f(49.0 * LOG2E) * (eq["r"][0] + eq["r"][1]))
        kern = float(max(npf * e1f + Ppf, qf * e2f + Pqf) + f(1.0e-3))
        out.append((p, inner, corner, two, max(G[p], G[q]), min(corner, lane), kern))
    return np.array(out)


def window(kind, rng, n):
    if kind == "noise":
        return np.rint(rng.normal(0, 32, n))
    if kind == "offset":
        return np.rint(rng.normal(3000, 32, n))
    if kind == "step":
        s = rng.integers(200, n - 200)
        y = rng.normal(0, 32, n); y[s:] += rng.choice([40, 200, 2000])
        return np.rint(y)
    if kind == "spikes":
        y = rng.normal(0, 32, n); idx = rng.integers(0, n, 6); y[idx] += rng.normal(0, 2000, 6)
        return np.rint(y)
    if kind == "ramp":
        return np.rint(rng.normal(0, 32, n) + np.linspace(0, 300, n))
    if kind == "quiet":
        y = rng.normal(0, 1.0, n); y[n // 3: n // 3 + 64] = rng.normal(0, 200, 64)
        return np.rint(y)
    raise ValueError(kind)


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(5)
    thr = 18.4204807339517 * LOG2E                         # the bench's min_gain in log2 units
    worst = 0.0
    worst2 = 0.0
    for kind in ("noise", "offset", "step", "spikes", "ramp", "quiet"):
        loose_c, loose_t, loose_r, kept_c, kept_t, kept_r, nblk = [], [], [], 0, 0, 0, 0
        for t in range(trials):
            n = int(rng.choice([800, 3000, 10000]))
            b = bounds(window(kind, rng, n))
            ok = np.isfinite(b[:, 1])
            b = b[ok]
            viol = (b[:, 1] - b[:, 3]).max()               # interior gain above the two-boundary bound?
            worst = max(worst, viol)
            assert (b[:, 1] <= b[:, 2] + 1e-6).all(), "corner bound violated"
            assert viol <= 1e-6, "two-boundary bound violated by %g (%s, n = %d)" % (viol, kind, n)
            viol2 = (b[:, 1] - b[:, 5]).max()
            worst2 = max(worst2, viol2)
            assert viol2 <= 1e-6, "register form of the two-boundary bound violated by %g (%s, n = %d)" % (viol2, kind, n)
            loose_r += list(b[:, 5] - b[:, 4])
            # float32 form: the screened gains it is compared with are themselves within delta = 0.02 + 8e-6 n of the exact ones
            viol3 = (b[:, 1] - b[:, 6]).max()
            assert viol3 <= 0.02 + 8e-6 * n, "float32 form of the bound violated by %g (%s, n = %d)" % (viol3, kind, n)
            loose_c += list(b[:, 2] - b[:, 4]); loose_t += list(b[:, 3] - b[:, 4])
            cand = (b[:, 0] >= 104) & (b[:, 0] <= n - 108)
            kept_c += int((b[cand, 2] >= thr - 0.4).sum()); kept_t += int((b[cand, 3] >= thr - 0.4).sum()); nblk += int(cand.sum())
            kept_r += int((b[cand, 5] >= thr - 0.4).sum())
        lc, lt, lr = np.array(loose_c), np.array(loose_t), np.array(loose_r)
        print("%-7s register form (min with the corner bound): median %6.2f p99 %7.2f, blocks kept %5.2f %%" % (kind, np.median(lr), np.percentile(lr, 99), 100.0 * kept_r / nblk))
        print("%-7s bound minus the larger boundary gain: corner median %6.2f p99 %7.2f | two boundaries median %6.2f p99 %7.2f | "
              "blocks kept at the bench threshold: %5.2f %% / %5.2f %%" % (kind, np.median(lc), np.percentile(lc, 99), np.median(lt),
                                                                          np.percentile(lt, 99), 100.0 * kept_c / nblk, 100.0 * kept_t / nblk))
    print("largest (interior gain - two-boundary bound) seen: %.3g, register form: %.3g (must be <= 0)" % (worst, worst2))


if __name__ == "__main__":
    main()
