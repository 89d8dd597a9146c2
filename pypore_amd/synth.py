"""Bit-reproducible synthetic ionic-current traces (integer-only generator).

SURVEY.md section 8(d) fixes the signal model used by every parity test, golden
vector and bench line: samples live on an ADC grid of 2**-5 pA, levels cycle
through [50, 42, 55, 38, 47] pA, and the noise is an Irwin-Hall(4) sum of the
four 16-bit lanes of splitmix64, centred and rescaled to sigma = 32 counts
(1 pA).  Everything is integer arithmetic on uint64 so numpy (here) and the HIP
generator kernel (csrc/synth.hip) produce identical counts on any version.

This module is host-side test/bench infrastructure; it is not on the product
hot path.
"""
import numpy as np

QUANTUM = 2.0 ** -5                      # pA per ADC count
LEVEL_COUNTS = np.array([1600, 1344, 1760, 1216, 1504], dtype=np.int64)  # 50,42,55,38,47 pA
OPEN_COUNTS = 3520                       # 110 pA open-channel level (config 3)
GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
NOISE_MUL = 887                          # 32 / sqrt(4*(65536**2-1)/12) * 2**20, rounded
NOISE_SHIFT = 20
DWELL_SEED_XOR = np.uint64(0xD1B54A32D192ED03)


def splitmix64(z):
    """Finaliser of splitmix64 applied to uint64 array `z` (wrapping arithmetic)."""
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def noise_sum(seed, start, n):
    """Sum of the four 16-bit lanes of the sample hash for indices [start, start+n): Irwin-Hall(4), 0 .. 262140."""
    idx = np.arange(start + 1, start + n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = splitmix64(np.uint64(seed) + idx * GOLDEN)
    return ((h & np.uint64(0xFFFF)) + ((h >> np.uint64(16)) & np.uint64(0xFFFF))
            + ((h >> np.uint64(32)) & np.uint64(0xFFFF)) + (h >> np.uint64(48))).astype(np.int64)


def noise_counts(seed, start, n):
    """Integer noise (sigma ~= 32 counts) for sample indices [start, start+n)."""
    s = noise_sum(seed, start, n)
    return ((s - 131070) * NOISE_MUL + (1 << (NOISE_SHIFT - 1))) >> NOISE_SHIFT


def dwell_table(seed, total, lo=1000, hi=20000):
    """Dwell lengths d_k = lo + splitmix64(seed^X + (k+1)*G) % (hi-lo), enough to cover `total`."""
    out = []
    acc = 0
    k = 0
    base = np.uint64(seed) ^ DWELL_SEED_XOR
    while acc < total:
        m = 4096
        kk = np.arange(k + 1, k + m + 1, dtype=np.uint64)
        with np.errstate(over="ignore"):
            d = (splitmix64(base + kk * GOLDEN) % np.uint64(hi - lo)).astype(np.int64) + lo
        out.append(d)
        acc += int(d.sum())
        k += m
    d = np.concatenate(out)
    ends = np.cumsum(d)
    nseg = int(np.searchsorted(ends, total, side="left")) + 1
    return d[:nseg]


def step_counts(n, dwell, seed, level_offset=0):
    """Fixed-dwell step signal: level k = LEVEL_COUNTS[(i // dwell + level_offset) % 5] + noise."""
    seg = (np.arange(n, dtype=np.int64) // dwell + level_offset) % 5
    return LEVEL_COUNTS[seg] + noise_counts(seed, 0, n)


def random_dwell_counts(n, seed, lo=1000, hi=20000, chunk=1 << 24):
    """Random-dwell step signal, dwell ~ U[lo, hi) from the hash stream (configs 3/5 interior)."""
    d = dwell_table(seed, n, lo, hi)
    ends = np.cumsum(d)
    out = np.empty(n, dtype=np.int32)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        seg = np.searchsorted(ends, np.arange(s, e, dtype=np.int64), side="right")
        out[s:e] = LEVEL_COUNTS[seg % 5] + noise_counts(seed, s, e - s)
    return out


def offgrid_trace(n, seed, sigma=1.0, lo=1000, hi=20000):
    """float64 pA on NO ADC grid (what a host-side filter, a resampler or np.random.normal test data produce): the
    level cycle of the step signals plus two independent, nearly Gaussian noise streams at incommensurate scales.  Only
    integer hashing and single correctly-rounded float64 operations: bit-reproducible on any IEEE machine."""
    d = dwell_table(seed, n, lo, hi)
    ends = np.cumsum(d)
    seg = np.searchsorted(ends, np.arange(n, dtype=np.int64), side="right")
    level = (LEVEL_COUNTS[seg % 5].astype(np.float64)) * QUANTUM
    a = noise_sum(seed, 0, n).astype(np.float64)                # Irwin-Hall(4) of 16-bit lanes, mean 131070, sd 37837.23
    b = noise_sum(seed ^ 0x5DEECE66D, 0, n).astype(np.float64)
    x = (a - 131070.0) / 37837.227
    x = x + ((b - 131070.0) / 37837.227) * 0.0031415926535897933
    return level + x * float(sigma)


def palindrome_counts(n, a, seed):
    """Levels A | B | A with steps at a and n - a and palindromic noise (sample i == sample n-1-i): the two steps are
    exactly tied candidates of a window that spans the event."""
    i = np.arange(n, dtype=np.int64)
    level = np.where((i >= a) & (i < n - a), LEVEL_COUNTS[1], LEVEL_COUNTS[0])
    noise = noise_counts(seed, 0, (n + 1) // 2)
    return (level + noise[np.minimum(i, n - 1 - i)]).astype(np.int32)


def counts_to_pa(counts, dtype=np.float32):
    """ADC counts -> pA on the 2**-5 grid (exact in float32 and float64)."""
    return (np.asarray(counts).astype(np.float64) * QUANTUM).astype(dtype)


def config1(dtype=np.float64):
    """BASELINE config 1: 5 levels x 2000 samples, seed 1."""
    return counts_to_pa(step_counts(10000, 2000, 1), dtype)


def config2_event(ev, n=50000, dwell=10000, dtype=np.float64):
    """BASELINE config 2, event `ev`: 5 levels x 10000 samples, seed = event id."""
    return counts_to_pa(step_counts(n, dwell, ev), dtype)


def file_trace_table(n, seed, gap=50000, ev_lo=150000, ev_hi=1000000, lo=1000, hi=20000):
    """Segment table (exclusive ends, level counts) of the config-3 trace of file_trace_counts, for the
    device generator (ps_synth_trace), plus the list of (start, length) of the blockade events."""
    seg_end, level, events = [], [], []
    pos = 0
    k = 0
    base = np.uint64(seed) ^ np.uint64(0xA0761D6478BD642F)
    while pos < n:
        g = min(gap, n - pos)
        pos += g
        seg_end.append(pos); level.append(OPEN_COUNTS)
        if pos >= n:
            break
        with np.errstate(over="ignore"):
            ln = int(splitmix64(base + np.uint64(k + 1) * GOLDEN) % np.uint64(ev_hi - ev_lo)) + ev_lo
        k += 1
        ln = min(ln, n - pos)
        d = dwell_table(seed + 7919 * k, ln, lo, hi)
        ends = np.minimum(np.cumsum(d), ln)
        lv = LEVEL_COUNTS[(np.arange(len(d)) + k) % 5]
        seg_end.extend((pos + ends).tolist()); level.extend(lv.tolist())
        if pos + ln < n:
            events.append((pos, ln))
        pos += ln
    return np.array(seg_end, dtype=np.int64), np.array(level, dtype=np.int32), events


def file_trace_counts(n, seed, gap=50000, ev_lo=150000, ev_hi=1000000, lo=1000, hi=20000):
    """BASELINE config 3 trace: open channel (110 pA) for `gap` samples between events,
    events ev_lo..ev_hi samples long with interior dwells U[lo,hi).  Returns (counts int32,
    list of (start, length) of the blockade events that lie fully inside the trace)."""
    out = np.empty(n, dtype=np.int32)
    events = []
    pos = 0
    k = 0
    base = np.uint64(seed) ^ np.uint64(0xA0761D6478BD642F)
    while pos < n:
        g = min(gap, n - pos)
        out[pos:pos + g] = OPEN_COUNTS + noise_counts(seed, pos, g)
        pos += g
        if pos >= n:
            break
        with np.errstate(over="ignore"):
            ln = int(splitmix64(base + np.uint64(k + 1) * GOLDEN) % np.uint64(ev_hi - ev_lo)) + ev_lo
        k += 1
        ln = min(ln, n - pos)
        d = dwell_table(seed + 7919 * k, ln, lo, hi)
        ends = np.cumsum(d)
        seg = np.searchsorted(ends, np.arange(ln, dtype=np.int64), side="right")
        out[pos:pos + ln] = LEVEL_COUNTS[(seg + k) % 5] + noise_counts(seed, pos, ln)
        if pos + ln < n:
            events.append((pos, ln))
        pos += ln
    return out, events
