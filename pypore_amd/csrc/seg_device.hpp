// seg_device.hpp -- device side of libporeseg: window scan, recursion drivers, kernels.
//
// gfx950 only (wave64, 256 CUs).  One workgroup of NT threads works on one job; all control
// flow of the recursion is workgroup-uniform (decisions are broadcast through LDS), the
// per-candidate work is spread over the lanes.
//
// Reference functions restated here (PyPore/cparsers.pyx):
//   var_c                 :31-38    -> ref_var()
//   _best_split_stepwise  :157-178  -> scan_window()
//   _recursive_split      :180-203  -> find_split() + spine_kernel / tree_kernel
//   _best_single_split    :134-155  -> single_scan_kernel (mode 1)
//   _best_split_stepwise_score :222-249 -> single_scan_kernel (mode 0)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "poreseg.h"

namespace ps {

constexpr int NT = 256;          // threads per workgroup: 4 waves, one per SIMD of a CU
constexpr int NWAVE = NT / 64;
constexpr int LDS_STACK = 128;   // DFS stack entries kept in LDS before spilling to HBM

enum : int { KIND_NONE = 0, KIND_HIT = 1, KIND_EARLY = 2, KIND_LATE = 3 };
enum : unsigned { ST_OFF_GRID = 1u, ST_OUT_OVERFLOW = 2u, ST_STACK_OVERFLOW = 4u };

struct DevCfg {
    const void *samples;
    int dtype, off_counts;
    float inv_q;
    double q, q2;
    int mw, maxw, W, half;
    double min_gain;
};

struct SpineJob {           // speculative spine of one tile: rec(start, end) without left subtrees
    int64_t base;           // offset of the event in the sample array
    int32_t start, end;     // chain anchor, event length
    int32_t stop;           // stop after the first spine anchor >= stop
    int32_t out_cap;
    int64_t out_off;        // into the private anchor scratch (int2 entries)
};

struct TreeJob {            // full in-order traversal of rec(start, end), first window index j0
    int64_t base;
    int32_t start, end, j0;
    int32_t out_cap;
    int64_t out_off;        // into the private boundary scratch (int32) and the spill stack (int2)
};

struct Shared {
    double wsum1[NWAVE], wsum2[NWAVE];
    double wbest[NWAVE];
    int widx[NWAVE];
    int bcast;
    int2 pop;
    int2 stack[LDS_STACK];
};

// ---- sample access --------------------------------------------------------------------------
__device__ __forceinline__ int load_count(const DevCfg &c, int64_t gi, unsigned &bad)
{
    if (c.dtype == PS_DTYPE_F32) {
        float x = static_cast<const float *>(c.samples)[gi];
        float k = x * c.inv_q;
        int ki = __float2int_rn(k);
        if (static_cast<float>(ki) != k || fabsf(k) >= 8388608.f) bad |= ST_OFF_GRID;
        return ki;
    }
    return static_cast<int>(static_cast<const int16_t *>(c.samples)[gi]) + c.off_counts;
}

// ---- exact (reference-order) arithmetic ----------------------------------------------------
// cparsers.pyx:31-38.  s1, s2 are EXACT integer sums of counts and counts^2 over the range
// (held in fp64), so dc and dc2 equal the reference's c[e-1]-c[s-1] and c2[e-1]-c2[s-1] bit
// for bit when quantum is a power of two.  No FMA contraction: the reference build has none.
__device__ __forceinline__ double ref_var(double s1, double s2, int n, double q, double q2)
{
#pragma clang fp contract(off)
    double dn = static_cast<double>(n);
    double dc = s1 * q, dc2 = s2 * q2;
    double m = dc / dn;
    double v = dc2 / dn;
    double mm = m * m;
    return v - mm;
}

__device__ __forceinline__ double ref_gain(double var_summed, int nl, double vl, int nr, double vr)
{
#pragma clang fp contract(off)
    double low = static_cast<double>(nl) * log(vl);
    double high = static_cast<double>(nr) * log(vr);
    double s = low + high;
    return var_summed - s;
}

// ---- workgroup primitives -------------------------------------------------------------------
// Exclusive prefix over the workgroup of two fp64 values that are exact integers (so the
// summation order does not matter); also returns the workgroup totals.
__device__ __forceinline__ void block_exscan2(double v1, double v2, double &e1, double &e2,
                                              double &t1, double &t2, Shared &sh)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double i1 = v1, i2 = v2;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double u1 = __shfl_up(i1, d), u2 = __shfl_up(i2, d);
        if (lane >= d) { i1 += u1; i2 += u2; }
    }
    if (lane == 63) { sh.wsum1[wave] = i1; sh.wsum2[wave] = i2; }
    __syncthreads();
    double b1 = 0, b2 = 0, s1 = 0, s2 = 0;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) {
        if (w < wave) { b1 += sh.wsum1[w]; b2 += sh.wsum2[w]; }
        s1 += sh.wsum1[w]; s2 += sh.wsum2[w];
    }
    e1 = b1 + i1 - v1; e2 = b2 + i2 - v2;
    t1 = s1; t2 = s2;
}

// a beats b: larger gain, or equal gain at the lower index (reference: strict '>' while
// ascending i => first maximum wins, cparsers.pyx:175-177).  idx -1 (no candidate above the
// threshold) carries the threshold itself and loses every tie as an unsigned index.
__device__ __forceinline__ bool beats(double ga, int ia, double gb, int ib)
{
    return ga > gb || (ga == gb && static_cast<unsigned>(ia) < static_cast<unsigned>(ib));
}

__device__ __forceinline__ int block_argmax(double g, int idx, Shared &sh)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        double og = __shfl_down(g, d);
        int oi = __shfl_down(idx, d);
        if (beats(og, oi, g, idx)) { g = og; idx = oi; }
    }
    if (lane == 0) { sh.wbest[wave] = g; sh.widx[wave] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double bg = sh.wbest[0]; int bi = sh.widx[0];
#pragma unroll
        for (int w = 1; w < NWAVE; ++w)
            if (beats(sh.wbest[w], sh.widx[w], bg, bi)) { bg = sh.wbest[w]; bi = sh.widx[w]; }
        sh.bcast = bi;
        sh.wbest[0] = bg;
    }
    __syncthreads();
    int r = sh.bcast;
    return r;
}

// ---- one window scan: cparsers.pyx:157-178 ------------------------------------------------------
// Window [ps, pe) of the event at `base`; candidates cand_lo..cand_hi (inclusive, event-local);
// returns the first index whose gain strictly exceeds every earlier gain and `thresh`, or -1.
// best_gain_out (thread 0 only, nullable) receives the winning gain (or thresh).
struct Work { long long windows, cands; };

__device__ int scan_window(const DevCfg &c, int64_t base, int ps, int pe, int cand_lo, int cand_hi,
                           double thresh, double *scores, Shared &sh, unsigned &bad, Work &wk,
                           double *best_gain_out = nullptr)
{
    const int n = pe - ps;
    const int ch = (n + NT - 1) / NT;
    const long long lo_ll = static_cast<long long>(threadIdx.x) * ch;
    const int lo = lo_ll < n ? static_cast<int>(lo_ll) : n;
    const int hi = (lo + ch < n) ? lo + ch : n;
    const int64_t g0 = base + ps;

    double s1 = 0, s2 = 0;
    for (int j = lo; j < hi; ++j) {
        double k = static_cast<double>(load_count(c, g0 + j, bad));
        s1 += k; s2 += k * k;
    }
    double a1, a2, t1, t2;
    block_exscan2(s1, s2, a1, a2, t1, t2, sh);

    const double var_summed = static_cast<double>(n) * log(ref_var(t1, t2, n, c.q, c.q2));
    double best = thresh;
    int bi = -1;
    for (int j = lo; j < hi; ++j) {
        const int i = ps + j;
        if (i >= cand_lo && i <= cand_hi) {
            const int nl = j, nr = n - j;
            double vl = ref_var(a1, a2, nl, c.q, c.q2);
            double vr = ref_var(t1 - a1, t2 - a2, nr, c.q, c.q2);
            double gain = ref_gain(var_summed, nl, vl, nr, vr);
            if (scores) scores[i] = gain;
            if (gain > best) { best = gain; bi = i; }
        }
        double k = static_cast<double>(load_count(c, g0 + j, bad));
        a1 += k; a2 += k * k;
    }
    int r = block_argmax(best, bi, sh);
    if (threadIdx.x == 0) {
        wk.windows += 1;
        wk.cands += (cand_hi >= cand_lo) ? (cand_hi - cand_lo + 1) : 0;
        if (best_gain_out) *best_gain_out = sh.wbest[0];
    }
    return r;
}

// ---- the window loop of _recursive_split: cparsers.pyx:186-201 ---------------------------------
// Windows j < j0 are known to hold no split (they were scanned with identical bounds by the
// parent frame, DESIGN.md "memoised left child").
__device__ int find_split(const DevCfg &c, int64_t base, int start, int end, int j0, int &kind,
                          Shared &sh, unsigned &bad, Work &wk)
{
    const long long lim = static_cast<long long>(end) - 2LL * c.mw;
    for (long long ps = static_cast<long long>(start) + static_cast<long long>(j0) * c.half; ps < lim;
         ps += c.half) {
        if (ps > static_cast<long long>(start) + c.maxw) {             // :189-191
            long long a = static_cast<long long>(start) + c.maxw, b = static_cast<long long>(end) - c.mw;
            kind = KIND_EARLY;
            return static_cast<int>(a < b ? a : b);
        }
        long long pe = ps + c.W;
        if (pe > end) pe = end;                                         // :193
        int s = -1;
        if (pe - ps > 2LL * c.mw)                                       // :164
            s = scan_window(c, base, static_cast<int>(ps), static_cast<int>(pe),
                            static_cast<int>(ps) + c.mw, static_cast<int>(pe) - c.mw,
                            c.min_gain, nullptr, sh, bad, wk);
        if (s >= 0) { kind = KIND_HIT; return s; }                      // :195-196
    }
    if (static_cast<long long>(end) - start <= c.maxw) { kind = KIND_NONE; return -1; }   // :199-200
    long long a = static_cast<long long>(start) + c.maxw, b = static_cast<long long>(end) - c.mw;
    kind = KIND_LATE;                                                   // :201
    return static_cast<int>(a < b ? a : b);
}

__device__ __forceinline__ int left_child_j0(int start, int split, int W, int half)
{
    long long d = static_cast<long long>(split) - W - start;
    return d < 0 ? 0 : static_cast<int>(d / half) + 1;
}

__device__ __forceinline__ void flush(unsigned bad, const Work &wk, unsigned *status, unsigned long long *work)
{
    if (bad) atomicOr(status, bad);
    if (threadIdx.x == 0) {
        atomicAdd(&work[0], static_cast<unsigned long long>(wk.windows));
        atomicAdd(&work[1], static_cast<unsigned long long>(wk.cands));
    }
}

// ---- phase 1: spine of rec(start, end), left subtrees skipped ------------------------------------
// out (private scratch, int2 = (anchor, kind)); meta[job] = (count, ended, dense position).
// After the chain stops, the workgroup reserves `count` slots in the dense list with one
// atomic and copies its anchors there so the host fetches a compact array.
__global__ __launch_bounds__(NT) void spine_kernel(DevCfg c, const SpineJob *jobs, int2 *scratch,
                                                   int2 *dense, int4 *meta, unsigned long long *dense_count,
                                                   unsigned *status, unsigned long long *work)
{
    __shared__ Shared sh;
    const SpineJob job = jobs[blockIdx.x];
    int2 *out = scratch + job.out_off;
    unsigned bad = 0;
    Work wk = {0, 0};
    int a = job.start, cnt = 0, ended = 0;
    for (;;) {
        int kind;
        int s = find_split(c, job.base, a, job.end, 0, kind, sh, bad, wk);
        if (kind == KIND_NONE) { ended = 1; break; }
        if (cnt < job.out_cap) { if (threadIdx.x == 0) out[cnt] = make_int2(s, kind); }
        else bad |= ST_OUT_OVERFLOW;
        ++cnt;
        a = s;
        if (a >= job.stop) break;
    }
    if (cnt > job.out_cap) cnt = job.out_cap;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long pos = atomicAdd(dense_count, static_cast<unsigned long long>(cnt));
        meta[blockIdx.x] = make_int4(cnt, ended, static_cast<int>(pos), 0);
        sh.bcast = static_cast<int>(pos);
    }
    __syncthreads();      // also orders thread 0's stores to `out` before the workgroup reads them
    const int pos = sh.bcast;
    for (int i = threadIdx.x; i < cnt; i += NT) dense[pos + i] = out[i];
    flush(bad, wk, status, work);
}

// ---- phase 3: in-order traversal of rec(start, end) -----------------------------------------------
__global__ __launch_bounds__(NT) void tree_kernel(DevCfg c, const TreeJob *jobs, int32_t *scratch,
                                                  int2 *spill, int32_t *counts, unsigned *status,
                                                  unsigned long long *work)
{
    __shared__ Shared sh;
    const TreeJob job = jobs[blockIdx.x];
    int32_t *out = scratch + job.out_off;
    int2 *sp_glob = spill + job.out_off;
    unsigned bad = 0;
    Work wk = {0, 0};
    int start = job.start, end = job.end, j0 = job.j0, sp = 0, cnt = 0;
    for (;;) {
        int kind;
        int s = find_split(c, job.base, start, end, j0, kind, sh, bad, wk);
        if (kind == KIND_NONE) {
            if (sp == 0) break;
            --sp;
            if (threadIdx.x == 0) sh.pop = sp < LDS_STACK ? sh.stack[sp] : sp_glob[sp - LDS_STACK];
            __syncthreads();
            const int2 top = sh.pop;
            __syncthreads();
            if (cnt < job.out_cap) { if (threadIdx.x == 0) out[cnt] = top.x; }
            else bad |= ST_OUT_OVERFLOW;
            ++cnt;
            start = top.x; end = top.y; j0 = 0;
            continue;
        }
        if (kind == KIND_EARLY) {                     // [split] + rec(split, end)
            if (cnt < job.out_cap) { if (threadIdx.x == 0) out[cnt] = s; }
            else bad |= ST_OUT_OVERFLOW;
            ++cnt;
            start = s; j0 = 0;
            continue;
        }
        // HIT / LATE: rec(start, s) first, then emit s and continue with rec(s, end)
        if (sp < LDS_STACK) {
            if (threadIdx.x == 0) sh.stack[sp] = make_int2(s, end);
        } else if (sp - LDS_STACK < job.out_cap) {
            if (threadIdx.x == 0) sp_glob[sp - LDS_STACK] = make_int2(s, end);
        } else {
            bad |= ST_STACK_OVERFLOW;
            break;
        }
        ++sp;
        __syncthreads();
        j0 = left_child_j0(start, s, c.W, c.half);
        end = s;
    }
    if (threadIdx.x == 0) counts[blockIdx.x] = cnt < job.out_cap ? cnt : job.out_cap;
    flush(bad, wk, status, work);
}

// ---- single scans for the API-completeness entry points -----------------------------------------
// mode 0: score_samples(no_split=True)  (window [0,n), candidates mw..n-mw, threshold min_gain)
// mode 1: best_single_split              (window [0,n-1), candidates 2..n-4, threshold 0)
__global__ __launch_bounds__(NT) void single_scan_kernel(DevCfg c, int n, int mode, double *scores,
                                                         double *gain_out, int *idx_out,
                                                         unsigned *status, unsigned long long *work)
{
    __shared__ Shared sh;
    unsigned bad = 0;
    Work wk = {0, 0};
    int r = -1;
    double g = 0.0;
    if (mode == 0) {
        if (n > 2 * c.mw)
            r = scan_window(c, 0, 0, n, c.mw, n - c.mw, c.min_gain, scores, sh, bad, wk, &g);
    } else {
        const int end = n - 1;
        if (end >= 1) r = scan_window(c, 0, 0, end, 2, end - 3, 0.0, nullptr, sh, bad, wk, &g);
    }
    if (threadIdx.x == 0) { *idx_out = r; *gain_out = g; }
    flush(bad, wk, status, work);
}

// ---- gather: final boundary list = for every true spine anchor: [its left subtree..., anchor] ----
struct Item { int32_t job; int32_t anchor; };

// single-workgroup exclusive scan of (tree count + 1) per item -> pos[n_items + 1]
__global__ __launch_bounds__(1024) void item_scan_kernel(const Item *items, const int32_t *counts,
                                                         int64_t n_items, int64_t *pos)
{
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t b0 = 0; b0 < n_items; b0 += 1024) {
        int64_t i = b0 + threadIdx.x;
        long long v = 0;
        if (i < n_items) v = 1 + (items[i].job >= 0 ? counts[items[i].job] : 0);
        long long inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            long long u = __shfl_up(inc, d);
            if (lane >= d) inc += u;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        long long basev = carry_s;
        for (int w = 0; w < wave; ++w) basev += wsum[w];
        if (i < n_items) pos[i] = basev + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = basev + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) pos[n_items] = carry_s;
}

__global__ __launch_bounds__(64) void gather_kernel(const Item *items, const TreeJob *jobs,
                                                    const int32_t *counts, const int32_t *scratch,
                                                    const int64_t *pos, int64_t n_items,
                                                    int32_t *bounds, int64_t cap)
{
    const int64_t it = blockIdx.x;
    if (it >= n_items) return;
    const Item item = items[it];
    const int64_t p = pos[it];
    int cnt = 0;
    if (item.job >= 0) {
        cnt = counts[item.job];
        const int32_t *src = scratch + jobs[item.job].out_off;
        for (int i = threadIdx.x; i < cnt; i += 64)
            if (p + i < cap) bounds[p + i] = src[i];
    }
    if (threadIdx.x == 0 && p + cnt < cap) bounds[p + cnt] = item.anchor;
}

// bounds_off[e] = pos[first_item[e]]
__global__ void event_offsets_kernel(const int64_t *pos, const int64_t *first_item, int32_t n_ev,
                                     int64_t *bounds_off)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e <= n_ev) bounds_off[e] = pos[first_item[e]];
}

// ---- K2: per-segment statistics, core.py:209-223 ------------------------------------------------
// One workgroup per segment.  Sums of counts and counts^2 are exact (fp64 holds the integers),
// mean = q*S1/n, std = q*sqrt(S2/n - (S1/n)^2) (population), min/max exact.
__global__ __launch_bounds__(NT) void segstat_kernel(DevCfg c, const int64_t *ev_off, int32_t n_ev,
                                                     const int32_t *bounds, const int64_t *bounds_off,
                                                     ps_segstat *stats, unsigned *status)
{
    __shared__ Shared sh;
    __shared__ int smin[NWAVE], smax[NWAVE];
    const int64_t g = blockIdx.x;
    // event e: bounds_off[e] + e <= g < bounds_off[e+1] + e + 1
    int lo = 0, hi = n_ev - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (bounds_off[mid] + mid <= g) lo = mid; else hi = mid - 1;
    }
    const int e = lo;
    const int64_t boff = bounds_off[e];
    const int cnt = static_cast<int>(bounds_off[e + 1] - boff);
    const int s = static_cast<int>(g - boff - e);
    const int n_e = static_cast<int>(ev_off[e + 1] - ev_off[e]);
    const int a = s == 0 ? 0 : bounds[boff + s - 1];
    const int b = s == cnt ? n_e : bounds[boff + s];
    const int64_t g0 = ev_off[e];
    unsigned bad = 0;
    double s1 = 0, s2 = 0;
    int mn = 0x7fffffff, mx = static_cast<int>(0x80000000);
    for (int i = a + threadIdx.x; i < b; i += NT) {
        int k = load_count(c, g0 + i, bad);
        double d = static_cast<double>(k);
        s1 += d; s2 += d * d;
        mn = k < mn ? k : mn; mx = k > mx ? k : mx;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s1 += __shfl_down(s1, d); s2 += __shfl_down(s2, d);
        int om = __shfl_down(mn, d), ox = __shfl_down(mx, d);
        mn = om < mn ? om : mn; mx = ox > mx ? ox : mx;
    }
    if (lane == 0) { sh.wsum1[wave] = s1; sh.wsum2[wave] = s2; smin[wave] = mn; smax[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t1 = 0, t2 = 0;
        int tm = 0x7fffffff, tx = static_cast<int>(0x80000000);
        for (int w = 0; w < NWAVE; ++w) {
            t1 += sh.wsum1[w]; t2 += sh.wsum2[w];
            tm = smin[w] < tm ? smin[w] : tm; tx = smax[w] > tx ? smax[w] : tx;
        }
        ps_segstat r;
        const int n = b - a;
        if (n > 0) {
            double m = t1 / n;
            double var = t2 / n - m * m;
            if (var < 0) var = 0;
            r.mean = m * c.q; r.std = sqrt(var) * c.q;
            r.min = tm * c.q; r.max = tx * c.q;
        } else {
            r.mean = r.std = r.min = r.max = __builtin_nan("");
        }
        stats[g] = r;
    }
    if (bad) atomicOr(status, bad);
}

// ---- synthetic trace generator (pypore_amd/synth.py twin) ---------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void synth_kernel(void *out, int dtype, int64_t n, unsigned long long seed,
                             const int64_t *seg_end, const int32_t *level, int64_t nseg)
{
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        int64_t lo = 0, hi = nseg - 1;                 // first segment with seg_end > i
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if (seg_end[mid] > i) hi = mid; else lo = mid + 1;
        }
        unsigned long long h = splitmix64(seed + static_cast<unsigned long long>(i + 1) * 0x9E3779B97F4A7C15ull);
        long long s = static_cast<long long>((h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48));
        long long noise = ((s - 131070) * 887 + (1 << 19)) >> 20;
        int k = level[lo] + static_cast<int>(noise);
        if (dtype == PS_DTYPE_F32) static_cast<float *>(out)[i] = static_cast<float>(k) * 0.03125f;
        else static_cast<int16_t *>(out)[i] = static_cast<int16_t>(k);
    }
}

}  // namespace ps
