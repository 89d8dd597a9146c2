// seg_bs.hpp -- block-sum window scan: one WAVE scans one window, no LDS image of the samples.
//
// K0 (blocksum_kernel) streams the trace once.  Per 8 samples it forms S1 = sum (k-m) and
// S2 = sum (k-m)^2 (m = first count of the event) and leaves their EXCLUSIVE prefix within each
// chunk of 256 blocks (16 B per block), plus the chunk totals.  The exact sums of [ps, J) at any
// block boundary J of a window are then one coalesced 16-byte load plus a per-chunk offset: a window
// scan needs no per-sample work for the blocks the bound discards (98 % of them).  Lane L of the wave
// takes boundaries L, L+64, ...: the boundary candidates are evaluated from the sums (centred per
// candidate on the rounded mean, so D = n*p2 - p1^2 never cancels), the monotone block bound of
// seg_device.hpp prunes with the neighbouring lane's values, and only surviving blocks read raw
// samples.  No barrier, no cross-wave reduction, ~10 KB of LDS: 8-16 such workgroups fit a CU.
//
// Included by seg_device.hpp after the common helpers (screen arithmetic, DPP primitives, scan_exact).
#pragma once

namespace ps {

constexpr int BS_WIDE = 23000;             // |k - m| must stay below this: 8 * BS_WIDE^2 < 2^32, n * BS_WIDE < 2^31
enum : unsigned { ST_WIDE_RANGE = 16u };
constexpr int BS_CHUNK = 256;              // blocks per K0 workgroup = extent of one prefix chunk
// Wide digest (DT & DT_WIDE): both moments as 64-bit integers -- (E1 lo, E1 hi, E2 lo, E2 hi) per block, the chunk
// totals alike.  |k - m| < BSW_LIM: a block's S2 < 2^49, a chunk prefix < 2^57, a window of 90 000 samples < 2^62.5.
// Made for events that were filtered and re-quantised on a fine grid (DataTypes.Event.parse: |count| < 2^22).
constexpr int BSW_LIM = 1 << 23;
template <int DT> constexpr bool bs_wide() { return (DT & DT_WIDE) != 0; }
__device__ __forceinline__ long long i64_of(int lo, int hi) { return (static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo); }
// int64 -> fp64, correctly rounded (hi * 2^32 and lo are exact, the fma rounds once); exact while |x| < 2^53
__device__ __forceinline__ double d_of_i64(long long x)
{
    return fma(static_cast<double>(static_cast<int>(x >> 32)), 4294967296.0, static_cast<double>(static_cast<unsigned>(x)));
}

__device__ __forceinline__ unsigned long long u64_of(unsigned lo, unsigned hi) { return (static_cast<unsigned long long>(hi) << 32) | lo; }

// ---- K0 -----------------------------------------------------------------------------------------------
// One thread per 8-sample block (global block index gb; event e owns blocks [ev_boff[e], ev_boff[e+1])).
// bs[gb] = (E1, -, E2 as fp64): sums of the blocks of gb's chunk that precede gb (one entry past the last
// block is written too: the end boundary of the last window).  chunk_tot[2*chunk] = (S1, -, S2 as fp64),
// chunk_tot[2*chunk+1] = (max |k|, max |k-m|, -, -).
// (the second moments are exact integers below 2^53 carried in fp64: the scan forms n*S2 - S1^2 there)
// A chunk may straddle events (different m): only differences inside one event are ever formed.
// Per-sample work is kept to the minimum the digest needs (the kernel is bound by vector-instruction issue as much
// as by HBM: the first version spent 39 VALU instructions per sample): fp32 samples are checked for integrality in
// float (x/q, v_rndne_f32, the difference OR-ed into one word), counts pass through min3 / max3 / add3 and one 24-bit
// multiply-add each; the range checks (|k - m| < BS_WIDE, |k| < 2^23) are taken once per block from the block's
// min and max.  A workgroup that lies inside one event (the common case) reads the event's tables with scalar loads.
__device__ __forceinline__ int wave_incl_scan_i32(int x)
{
#define PS_STEP(CTRL, RM) { x += dpp_mov<CTRL, RM>(0, x); }
    PS_DPP_STEPS(PS_STEP)
#undef PS_STEP
    return x;
}
__device__ __forceinline__ double wave_incl_scan_f64(double x)
{
#define PS_STEP(CTRL, RM) { x += dpp_movd<CTRL, RM>(x); }
    PS_DPP_STEPS(PS_STEP)
#undef PS_STEP
    return x;
}
__device__ __forceinline__ long long wave_incl_scan_i64(long long x)
{
#define PS_STEP(CTRL, RM) { const int lo_ = dpp_mov<CTRL, RM>(0, static_cast<int>(x)), hi_ = dpp_mov<CTRL, RM>(0, static_cast<int>(x >> 32)); \
                            x += i64_of(lo_, hi_); }
    PS_DPP_STEPS(PS_STEP)
#undef PS_STEP
    return x;
}
__device__ __forceinline__ int wave_max_i32(int x)            // result in lane 63
{
#define PS_STEP(CTRL, RM) { x = max(x, dpp_mov<CTRL, RM>(static_cast<int>(0x80000000), x)); }
    PS_DPP_STEPS(PS_STEP)
#undef PS_STEP
    return x;
}

template <int DT>
__global__ __launch_bounds__(256) void blocksum_kernel(DevCfg c, const int64_t *ev_start, const int64_t *ev_len,
                                                       const int64_t *ev_boff, int n_ev, int64_t n_samples, int4 *bs,
                                                       int4 *ev_info, int4 *chunk_tot, unsigned *status)
{
    constexpr bool WIDE = bs_wide<DT>();
    __shared__ double w2[4];
    __shared__ long long wl1[4], wl2[4];               // (wide digest)
    __shared__ int w1[4], smax[4], symax[4];
    const long long wg0 = blockIdx.x * 256LL;
    const long long gb = wg0 + threadIdx.x;
    const long long nb_total = ev_boff[n_ev];
    unsigned bad = 0;
    int mabs = 0, yabs = 0;                        // max |k| over the block's samples, max |k - m|
    int s1 = 0;
    unsigned s2 = 0;
    unsigned long long s2w = 0;                        // (wide digest: 8 * 2^46)
    // event of the workgroup's first block (uniform search)
    int e_first = 0;
    {
        const long long gfirst = min(wg0, nb_total - 1);
        int lo = 0, hi = n_ev - 1;                 // event e: ev_boff[e] <= gb < ev_boff[e+1]
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (ev_boff[mid] <= gfirst) lo = mid; else hi = mid - 1;
        }
        e_first = lo;
    }
    const bool one_event = wg0 + 255 < ev_boff[e_first + 1];     // the whole workgroup lies in event e_first (uniform)
    if (gb < nb_total) {
        int e = e_first;
        int64_t len, base;
        long long b;
        if (one_event) {                           // uniform indices: scalar loads
            len = ev_len[e_first]; base = ev_start[e_first]; b = gb - ev_boff[e_first];
        } else {
            while (ev_boff[e + 1] <= gb) ++e;      // (empty events are stepped over)
            len = ev_len[e]; base = ev_start[e]; b = gb - ev_boff[e];
        }
        const int64_t i0 = 8 * b;
        const int m = load_count<DT>(c, base, bad);
        int y[8];
        const int cnt = static_cast<int>(len - i0 < 8 ? len - i0 : 8);
        constexpr int ES = static_cast<int>(sizeof(typename Raw<DT>::type));
        const char *p = static_cast<const char *>(c.samples) + (base + i0) * ES;
        int ymin, ymax;                            // over the block's real samples
#define PS_MM8 { ymin = min(min(min(y[0], y[1]), min(y[2], y[3])), min(min(y[4], y[5]), min(y[6], y[7])));     \
                 ymax = max(max(max(y[0], y[1]), max(y[2], y[3])), max(max(y[4], y[5]), max(y[6], y[7]))); }
        if (cnt == 8 && (reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
            constexpr int NV = 8 * ES / 16;
            int4 raw[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) raw[v] = reinterpret_cast<const int4 *>(p)[v];
            if (sdt(DT) == PS_DTYPE_F32) {
                const f2 iq = {c.inv_q, c.inv_q};
                const float mf = static_cast<float>(m);                    // |m| < 2^23: exact
                const f2 mf2 = {mf, mf};
                unsigned nz = 0;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
#pragma clang fp contract(off)                                              // (x/q rounded first, as in to_count)
                    const int w[4] = {raw[v].x, raw[v].y, raw[v].z, raw[v].w};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f2 x = {__int_as_float(w[2 * h]), __int_as_float(w[2 * h + 1])};
                        const f2 t = x * iq;
                        const f2 r = {__builtin_rintf(t.x), __builtin_rintf(t.y)};
                        const f2 d = t - r;                                 // +0 exactly when t is an integer; NaN for inf/NaN
                        nz |= __float_as_uint(d.x) | __float_as_uint(d.y);
                        const f2 yf = r - mf2;                              // exact while |r| < 2^24 (else: saturates below -> wide)
                        y[v * 4 + 2 * h] = static_cast<int>(yf.x);
                        y[v * 4 + 2 * h + 1] = static_cast<int>(yf.y);
                    }
                }
                if (nz) bad |= ST_OFF_GRID;
                PS_MM8
            } else {
                const int w[4] = {raw[0].x, raw[0].y, raw[0].z, raw[0].w};
                const int om = c.off_counts - m;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    y[q] = ((q & 1) ? (w[q >> 1] >> 16) : static_cast<int>(static_cast<short>(w[q >> 1] & 0xffff))) + om;
                PS_MM8
            }
        } else if (sdt(DT) == PS_DTYPE_I16 && cnt == 8 && base + i0 >= 1 && base + i0 + 10 <= n_samples) {
            // int16 block that is not 16-byte aligned (events cut out of a file trace start anywhere):
            // dword loads, shifted by one sample when the block starts on an odd sample
            const uintptr_t a = reinterpret_cast<uintptr_t>(p);
            const int *q4 = reinterpret_cast<const int *>(a & ~static_cast<uintptr_t>(3));
            int v[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) v[q] = q4[q];            // (v[4] is inside the array: one more sample follows)
            const bool odd = (a & 2u) != 0;
            const int om = c.off_counts - m;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int w = odd ? static_cast<int>((static_cast<unsigned>(v[q]) >> 16) | (static_cast<unsigned>(v[q + 1]) << 16)) : v[q];
                y[2 * q] = static_cast<int>(static_cast<short>(w & 0xffff)) + om;
                y[2 * q + 1] = (w >> 16) + om;
            }
            PS_MM8
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) y[q] = q < cnt ? load_count<DT>(c, base + i0 + q, bad) - m : 0;   // padding = m: y = 0
            ymin = ymax = y[0];                    // (cnt >= 1: block b exists only if it holds a sample)
#pragma unroll
            for (int q = 1; q < 8; ++q)
                if (q < cnt) { ymin = min(ymin, y[q]); ymax = max(ymax, y[q]); }
        }
#undef PS_MM8
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            s1 += y[q];
            if constexpr (WIDE) s2w += static_cast<unsigned long long>(static_cast<long long>(y[q]) * static_cast<long long>(y[q]));
            else s2 += static_cast<unsigned>(__mul24(y[q], y[q]));         // |y| < BS_WIDE < 2^23, else the call is redone
        }
        constexpr int LIM = WIDE ? BSW_LIM : BS_WIDE;
        if (ymax >= LIM || ymin <= -LIM) bad |= ST_WIDE_RANGE;
        const int ka = m + ymin, kb = m + ymax;
        mabs = max(ka < 0 ? -ka : ka, kb < 0 ? -kb : kb);
        yabs = max(-ymin, ymax);
        if (sdt(DT) == PS_DTYPE_F32 && mabs >= 8388608) bad |= ST_OFF_GRID;     // |count| >= 2^23
        if (!WIDE && c.blk_mm) c.blk_mm[gb] = (ymin & 0xffff) | (ymax << 16);      // (|y| < BS_WIDE fits int16; otherwise the call is redone)
        if (b == 0) ev_info[e] = make_int4(m, 0, static_cast<int>(ev_boff[e] & 0xffffffffLL), static_cast<int>(ev_boff[e] >> 32));
    }
    // exclusive prefix over the workgroup: first moments in int32 (256 * 8 * BS_WIDE < 2^31), second moments as exact
    // integers in fp64
    mabs = wave_max_i32(mabs);
    yabs = wave_max_i32(yabs);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (WIDE) {
        const long long i1 = wave_incl_scan_i64(static_cast<long long>(s1));
        const long long i2 = wave_incl_scan_i64(static_cast<long long>(s2w));
        if (lane == 63) { wl1[wave] = i1; wl2[wave] = i2; smax[wave] = mabs; symax[wave] = yabs; }
        __syncthreads();
        long long o1 = 0, o2 = 0;
        for (int w = 0; w < wave; ++w) { o1 += wl1[w]; o2 += wl2[w]; }
        if (gb <= nb_total) {
            const long long e1 = o1 + i1 - s1, e2 = o2 + i2 - static_cast<long long>(s2w);
            bs[gb] = make_int4(static_cast<int>(e1), static_cast<int>(e1 >> 32), static_cast<int>(e2), static_cast<int>(e2 >> 32));
        }
        if (threadIdx.x == 255) {
            const long long t1 = o1 + i1, t2 = o2 + i2;
            chunk_tot[2 * blockIdx.x] = make_int4(static_cast<int>(t1), static_cast<int>(t1 >> 32), static_cast<int>(t2), static_cast<int>(t2 >> 32));
        }
    } else {
        const int i1 = wave_incl_scan_i32(s1);
        const double i2 = wave_incl_scan_f64(static_cast<double>(s2));
        if (lane == 63) { w1[wave] = i1; w2[wave] = i2; smax[wave] = mabs; symax[wave] = yabs; }
        __syncthreads();
        int o1 = 0;
        double o2 = 0.0;
        for (int w = 0; w < wave; ++w) { o1 += w1[w]; o2 += w2[w]; }
        if (gb <= nb_total) {
            const int e1 = o1 + i1 - s1;
            const double e2 = (o2 + i2) - static_cast<double>(s2);
            bs[gb] = make_int4(e1, 0, __double2loint(e2), __double2hiint(e2));
        }
        if (threadIdx.x == 255) {
            const double t2 = o2 + i2;
            chunk_tot[2 * blockIdx.x] = make_int4(o1 + i1, 0, __double2loint(t2), __double2hiint(t2));
        }
    }
    if (threadIdx.x == 255)
        chunk_tot[2 * blockIdx.x + 1] = make_int4(max(max(smax[0], smax[1]), max(smax[2], smax[3])),
                                                  max(max(symax[0], symax[1]), max(symax[2], symax[3])), 0, 0);
    if (bad) atomicOr(status, bad);
}

// ---- screen arithmetic from window-relative sums ------------------------------------------------------
struct BsEval { float g; f2 lg; f2 r; bool okL, okR; };
__device__ __forceinline__ double ent2(const int4 &v) { return __hiloint2double(v.w, v.z); }

// Screened gain of the split (nl | nr) from the exact sums about m of the left part (a1, a2) and the right
// part (b1, b2).  D = n*S2 - S1^2 is formed in fp64 (relative error kappa_m * 2^-52 with kappa_m = n*S2/D
// below 2^24 for |k-m| < BS_WIDE and a variance above the floor), so no re-centring and no conditioning
// guard are needed; everything after the conversion of D is fp32 (v_pk_*), as in seg_device.hpp.
__device__ __forceinline__ BsEval bs_eval(double a1d, double a2, double b1d, double b2, int nl, int nr, f2 cc, float vfloor)
{
    const double DL = fma(static_cast<double>(nl), a2, -(a1d * a1d));
    const double DR = fma(static_cast<double>(nr), b2, -(b1d * b1d));
    const f2 D = {static_cast<float>(DL), static_cast<float>(DR)};
    const f2 nv = {static_cast<float>(nl), static_cast<float>(nr)};
    const f2 r = {__builtin_amdgcn_rcpf(nv.x), __builtin_amdgcn_rcpf(nv.y)};
    const f2 u = D * r * r;                                               // variances (counts^2)
    const f2 lgu = {__builtin_amdgcn_logf(u.x), __builtin_amdgcn_logf(u.y)};
    BsEval o;
    o.lg = lgu - cc;
    o.r = r;
    o.okL = u.x >= vfloor;
    o.okR = u.y >= vfloor;
    const f2 t = nv * o.lg;
    o.g = -(t.x + t.y);
    return o;
}

struct BsQ { int j, a1; double a2; };                     // queued block (J-8, J): its end J, sums of [ps, J)
struct BsC { int j, a1; double a2; float g; int pad; };   // contender: candidate, its exact sums, screened gain
struct BsOff { int o1, pad; double o2; };                 // sums of the window's chunks before this one (+ the window constant)
// the same three for the wide digest (64-bit integer sums; |S1| < 2^40 shares a word with the window-relative position)
struct BsQW { long long ja, a2; };                        // ja = S1 * 2^18 + (J - ps)
struct BsCW { int j; float g; long long a1, a2; };
struct BsOffW { long long o1, o2; };
static_assert(sizeof(BsQW) == sizeof(BsQ) && sizeof(BsCW) == sizeof(BsC) && sizeof(BsOffW) == sizeof(BsOff), "LDS layout shared by both digests");
template <bool WIDE> struct BsTypes { typedef int s1_t; typedef double s2_t; typedef BsQ Q; typedef BsC C; typedef BsOff Off; };
template <> struct BsTypes<true> { typedef long long s1_t; typedef long long s2_t; typedef BsQW Q; typedef BsCW C; typedef BsOffW Off; };
// moments of a digest entry; conversions to the fp64 the screen computes in
template <bool WIDE> __device__ __forceinline__ typename BsTypes<WIDE>::s1_t bs_s1(const int4 &v)
{
    if constexpr (WIDE) return i64_of(v.x, v.y); else return v.x;
}
template <bool WIDE> __device__ __forceinline__ typename BsTypes<WIDE>::s2_t bs_s2(const int4 &v)
{
    if constexpr (WIDE) return i64_of(v.z, v.w); else return __hiloint2double(v.w, v.z);
}
__device__ __forceinline__ double bs_d(int x) { return static_cast<double>(x); }
__device__ __forceinline__ double bs_d(double x) { return x; }
__device__ __forceinline__ double bs_d(long long x) { return d_of_i64(x); }
__device__ __forceinline__ void bs_q_put(BsQ &q, int J, int ps, int a1, double a2) { q.j = J; q.a1 = a1; q.a2 = a2; }
__device__ __forceinline__ void bs_q_put(BsQW &q, int J, int ps, long long a1, long long a2) { q.ja = a1 * 262144LL + (J - ps); q.a2 = a2; }
__device__ __forceinline__ int bs_q_j(const BsQ &q, int ps) { return q.j; }
__device__ __forceinline__ int bs_q_j(const BsQW &q, int ps) { return ps + static_cast<int>(q.ja & 262143LL); }
__device__ __forceinline__ int bs_q_a1(const BsQ &q) { return q.a1; }
__device__ __forceinline__ long long bs_q_a1(const BsQW &q) { return q.ja >> 18; }
constexpr int BS_NC = 64;                                 // contenders kept per window
#ifndef PS_BS_G
#define PS_BS_G 5
#endif
constexpr int BS_G = PS_BS_G;                             // rows per group (loads in flight)
constexpr int BS_QN = 64 * BS_G + 64;                     // queued blocks; a drain is forced when a group may not fit
constexpr int BS_STRIDE = 63;                             // new boundaries per row (lane 0 repeats the previous row's last)
static_assert(sizeof(QEnt) * SharedT<64>::QN >= sizeof(BsQ) * BS_QN + sizeof(BsC) * BS_NC + 64 * 32 + 64 * sizeof(BsOff),
              "SharedT<64>::q too small");             // (64 * 32: staged blocks of the wide digest, 8 int32 each)

// Exact (reference-order, fp64) gain of one candidate from its exact integer sums about m.
__device__ __forceinline__ double bs_exact_gain(const DevCfg &c, int m, int a1, double a2, int T1, double T2, int nl, int n,
                                                double var_summed)
{
#pragma clang fp contract(off)
    // (the caller guarantees n * max|k|^2 < 2^53: every product and sum below is then an exact integer in fp64)
    const double dm = static_cast<double>(m);
    // uncentred sums: sum k = a1 + nl*m ; sum k^2 = a2 + 2*m*a1 + nl*m^2   (exact integers below 2^53)
    const double l1 = static_cast<double>(a1) + static_cast<double>(nl) * dm;
    const double l2 = a2 + 2.0 * dm * static_cast<double>(a1) + static_cast<double>(nl) * dm * dm;
    const int nr = n - nl;
    const int b1 = T1 - a1;
    const double b2 = T2 - a2;
    const double r1 = static_cast<double>(b1) + static_cast<double>(nr) * dm;
    const double r2 = b2 + 2.0 * dm * static_cast<double>(b1) + static_cast<double>(nr) * dm * dm;
    const double vl = ref_var(l1, l2, nl, c.q, c.q2), vr = ref_var(r1, r2, nr, c.q, c.q2);
    return ref_gain(var_summed, nl, vl, nr, vr);
}

__device__ __forceinline__ int lanes_below(unsigned long long mask)      // set bits of `mask` below this lane
{
    return __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mask), 0u));
}
// Value of lane-1 (wave_shr:1, the GFX9 whole-wave shift); lane 0 keeps its own.
__device__ __forceinline__ int from_lane_below(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ float from_lane_below(float x) { return __int_as_float(from_lane_below(__float_as_int(x))); }
// Value of lane+1 (wave_shl:1); lane 63 keeps its own.
__device__ __forceinline__ int from_lane_above(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ float from_lane_above(float x) { return __int_as_float(from_lane_above(__float_as_int(x))); }

// One wave scans the window [ps, pe) of event `ev` (samples at c.samples[base + .]).
//
// Phase 0 (every window): in row r lane L takes block boundary t = 63 r + L (J = g0 + 8t; lane 0 repeats
// the previous row's last boundary so that every block finds its left neighbour one lane below): sums of
// [ps, J) from the K0 prefix, boundary candidate evaluated, the block (J-8, J) bounded with the lane
// below's left-side values and queued in LDS if it survives (slots from a ballot, no atomics); the queue
// is drained -- interior candidates evaluated from raw samples -- when it may overflow and at the end.
// Top-2 over the wave decides.
// Phase 1 (ambiguous windows only, ~1 %): the same sweep with the final maximum known collects the
// contenders (screened gain within 3 delta of the decision level) and the reference's fp64 arithmetic
// picks among them.  A whole-window fp64 scan remains for guard failures and contender overflow.
template <int DT, bool ROWSKIP = true>
__device__ int scan_window_bs(const DevCfg &c, int ev, int64_t base, int ps, int pe, int cand_lo, int cand_hi,
                              double thresh, SharedT<64> &sh, unsigned &bad, Work &wk)
{
    constexpr bool WIDE = bs_wide<DT>();
    typedef typename BsTypes<WIDE>::s1_t s1_t;         // first / second moments about m: int32 / fp64 (exact integers), or
    typedef typename BsTypes<WIDE>::s2_t s2_t;         // both int64 with the wide digest
    typedef typename BsTypes<WIDE>::Q BsQ_t;
    typedef typename BsTypes<WIDE>::C BsC_t;
    typedef typename BsTypes<WIDE>::Off BsOff_t;
    const int lane = threadIdx.x & 63;                 // (one wave; possibly one of several in its workgroup)
    const int n = pe - ps;
    const int g0 = (ps + 7) & ~7, g1 = pe & ~7;
    const int nblk = (g1 - g0) >> 3;
    if (nblk < 4 || n > 90000 || c.mode == MODE_EXACT || cand_lo < g0 || cand_hi > g1) {
        // tiny window, or candidates in the ragged head/tail (min_width < 8): exact scan straight from HBM
        if (lane == 0) wk.exact += 1;
        return scan_exact<64, DT>(c, nullptr, base + ps, ps, n, cand_lo, cand_hi, thresh, nullptr, sh, bad, nullptr);
    }
    const int4 info = c.ev_info[ev];
    const int m = info.x;
    const long long gb0 = ((static_cast<long long>(static_cast<unsigned>(info.w)) << 32) | static_cast<unsigned>(info.z)) + (g0 >> 3);
    const int4 *bsw = c.bsum + gb0;                                    // bsw[t]: chunk prefix at boundary t = 0..nblk
    const int c0 = static_cast<int>(gb0 >> 8), nch = static_cast<int>((gb0 + nblk) >> 8) - c0 + 1;     // chunks touched (<= 45)
    const int gbl = static_cast<int>(gb0 & 255);                       // chunk of boundary t: (gbl + t) >> 8
    PS_STAMP_AT(wk, 5);                                // (diagnostic) entry, event info
    // everything the window needs before its first boundary, issued together
    const int nh = g0 - ps, nt = pe - g1;              // ragged head [ps, g0) and tail [g1, pe): <= 7 raw samples each
    int yht = 0;
    if (lane < nh) yht = load_count<DT>(c, base + ps + lane, bad) - m;
    if (lane >= 32 && lane - 32 < nt) yht = load_count<DT>(c, base + g1 + (lane - 32), bad) - m;
    int4 ct = make_int4(0, 0, 0, 0), cm = make_int4(0, 0, 0, 0);
    if (lane < nch) { ct = c.chunk_tot[2 * (c0 + lane)]; cm = c.chunk_tot[2 * (c0 + lane) + 1]; }
    const int4 e0 = bsw[0], eN = bsw[nblk];
    const int nbnd = nblk + 1;
    const int rows = (nbnd - 1 + BS_STRIDE - 1) / BS_STRIDE;           // nbnd >= 5
    const int4 row0 = bsw[min(lane, nblk)];
    const int tS = min(nblk, lane * rows);             // one sampled boundary per lane, spread over the window
    const int4 smp = bsw[tS];
    // head/tail sums (one scan: lanes 0..31 head, 32..63 tail), chunk offsets (exclusive scan of the totals)
    double hs1 = static_cast<double>(yht), hs2 = static_cast<double>(yht) * static_cast<double>(yht);
    wave_incl_scan2(hs1, hs2);
    const double H1d = __shfl(hs1, 31), H2d = __shfl(hs2, 31);
    const double TL1d = __shfl(hs1, 63) - H1d, TL2d = __shfl(hs2, 63) - H2d;
    const s1_t cs1 = bs_s1<WIDE>(ct);
    const s2_t cs2 = bs_s2<WIDE>(ct);
    s1_t cx1;                                          // sums of the chunks before this lane's
    s2_t cx2;
    if constexpr (WIDE) {
        cx1 = wave_incl_scan_i64(cs1) - cs1;
        cx2 = wave_incl_scan_i64(cs2) - cs2;
    } else {
        double ci1 = static_cast<double>(cs1), ci2 = cs2;
        wave_incl_scan2(ci1, ci2);
        cx1 = static_cast<int>(ci1 - static_cast<double>(cs1));
        cx2 = ci2 - cs2;
    }
    int mabs = cm.x, nymax = -cm.y;                     // max |k| and max |k-m| over the chunks touched
    wave_minmax(nymax, mabs);                          // (the min slot carries -max |k-m|)
    mabs = __shfl(mabs, 63);
    const float ymaxf = static_cast<float>(-__shfl(nymax, 63));
    BsQ_t *queue = reinterpret_cast<BsQ_t *>(sh.q);
    BsC_t *cont = reinterpret_cast<BsC_t *>(queue + BS_QN);
    int4 *ybuf = reinterpret_cast<int4 *>(cont + BS_NC);               // 64 staged blocks: 8 int16 offsets (wide digest: 8 int32)
    BsOff_t *coff = reinterpret_cast<BsOff_t *>(ybuf + 128);
    // a(t) = E[t] + off[chunk(t)] : sums of [ps, g0 + 8t) about m  (off includes the head and -E[0])
    // (head and tail: at most 7 samples each, their sums are small exact integers in fp64)
    const s1_t K1 = static_cast<s1_t>(H1d) - bs_s1<WIDE>(e0);
    const s2_t K2 = static_cast<s2_t>(H2d) - bs_s2<WIDE>(e0);
    ps_sync<64>();                                  // previous user of sh.q (this wave) is done
    {
        BsOff_t o;
        if constexpr (!WIDE) o.pad = 0;
        o.o1 = cx1 + K1; o.o2 = cx2 + K2;
        coff[lane] = o;
    }
    ps_sync<64>();
    const BsOff_t oN = coff[nch - 1];
    const s1_t T1 = bs_s1<WIDE>(eN) + oN.o1 + static_cast<s1_t>(TL1d);
    const s2_t T2 = bs_s2<WIDE>(eN) + oN.o2 + static_cast<s2_t>(TL2d);     // window totals about m
    const double T1d = bs_d(T1), T2d = bs_d(T2);
    const double dn = static_cast<double>(n);
    const double Dtot = dn * T2d - T1d * T1d;
    if (!(Dtot > 0.0)) {
        if (lane == 0) wk.exact += 1;
        return scan_exact<64, DT>(c, nullptr, base + ps, ps, n, cand_lo, cand_hi, thresh, nullptr, sh, bad, nullptr);
    }
    const float rn = __builtin_amdgcn_rcpf(static_cast<float>(n));
    const float c0f = __builtin_amdgcn_logf(static_cast<float>(Dtot) * rn * rn);
    const f2 cc = {c0f, c0f};
    const float dlt = screen_delta_log2(n);
    const float thr_log2 = static_cast<float>(thresh * 1.4426950408889634);
    const float dthr = dlt + 3.0e-6f * static_cast<float>(n) + 1.0e-6f * fabsf(thr_log2);
    const float nf = static_cast<float>(n);
    const float LOG2E = 1.4426950408889634f;
    // variance floor: the reference's own fp64 rounding (mabs^2 * 2^-52 * n) and the fp64 D above (kappa_m <= 2^26)
    // (wide digest: S2 is rounded to fp64 once, kappa_m <= 2^22 keeps D to 2^-30)
    const float vfloor = fmaxf(static_cast<float>(mabs) * static_cast<float>(mabs) * 1.0e-9f, ymaxf * ymaxf * (WIDE ? 2.4e-7f : 1.5e-8f));
    const unsigned crange = static_cast<unsigned>(cand_hi - cand_lo);

    int result = -2;
    bool anyflag = false;
    float Tprune, Tc = INFINITY;
    float cbound = INFINITY;                           // bound of the stretch between this lane's sample and the next lane's
    bool hitlike = false;                              // (uniform) a sampled candidate lies above the threshold band
    {
        // pruning level from the sampled boundary candidates
        const int J = g0 + 8 * tS;
        const BsOff_t off = coff[(gbl + tS) >> 8];
        const s1_t a1 = bs_s1<WIDE>(smp) + off.o1;
        const s2_t a2 = bs_s2<WIDE>(smp) + off.o2;
        const BsEval e = bs_eval(bs_d(a1), bs_d(a2), bs_d(static_cast<s1_t>(T1 - a1)), bs_d(static_cast<s2_t>(T2 - a2)),
                                 max(J - ps, 1), max(pe - J, 1), cc, vfloor);
        const bool inr = static_cast<unsigned>(J - cand_lo) <= crange;
        const float bmine = (inr && e.okL && e.okR) ? e.g : -INFINITY;
        float bm = bmine;
#define PS_STEP(CTRL, RM) { bm = fmaxf(bm, dpp_movf<CTRL, RM>(-INFINITY, bm)); }
        PS_DPP_STEPS(PS_STEP)
#undef PS_STEP
        bm = __shfl(bm, 63);
        Tprune = fmaxf(thr_log2 - dthr, bm - 2.0f * dlt) - 2.0f * dlt;
        // A window that holds a split (a sampled gain above the threshold band; rows <= 64: windows up to 32 000
        // samples): the same monotone bound as for an 8-sample block, applied to the whole stretch [J, Jb) up to the
        // NEXT lane's sample (left side from this sample, right side from that one; loose by about Jb - J nats, nothing
        // next to the thousands of nats of a real step).  Nearly every stretch then lies below the pruning level, and
        // the sweep skips the rows that lie in such stretches altogether, loads included: a candidate there is provably
        // more than 2 delta below the winner.  Windows without such a sample (all subtree windows, half of the spine's)
        // skip this and sweep every row: for them the bookkeeping would only cost (measured: +1.6 us per window).
        hitlike = ROWSKIP && c.prune && rows <= 64 && bm > thr_log2 + dthr;
        if (hitlike) {
            const int Jb = from_lane_above(J);
            const float bRb = from_lane_above(e.lg.y), rreb = from_lane_above(e.r.y);
            const bool okRb = from_lane_above(static_cast<int>(e.okR)) != 0;
            const int B = Jb - J, nla = J - ps, nrb = pe - Jb;
            if (B <= 0) cbound = inr ? bmine == -INFINITY ? INFINITY : bmine : -INFINITY;     // (clamped lanes: the last boundary itself)
            else {
                const float Bf = static_cast<float>(B), nlaf = static_cast<float>(nla), nrbf = static_cast<float>(nrb);
                const float h0 = -fmaf(nlaf, e.lg.x, (nrbf + Bf) * (bRb - Bf * LOG2E * rreb));
                const float h1 = -fmaf(nlaf + Bf - 1.0f, e.lg.x - (Bf - 1.0f) * LOG2E * e.r.x, (nrbf + 1.0f) * (bRb - LOG2E * rreb));
                cbound = (e.okL && okRb && nla >= 1 && nrb >= 1) ? fmaxf(h0, h1) : INFINITY;
            }
        }
    }
    PS_STAMP_AT(wk, 0);                                // loads, totals, wave scans, pruning level
    int ccount = 0;
#define PS_COLLECT(COND, G, JJ, A1, A2)                                                                       \
    {                                                                                                         \
        const bool cp_ = (COND);                                                                              \
        const unsigned long long cm_ = __ballot(cp_);                                                         \
        if (cm_) {                                                                                            \
            const int cs_ = ccount + lanes_below(cm_);                                                        \
            if (cp_ && cs_ < BS_NC) {                                                                         \
                BsC_t e_; e_.j = (JJ); e_.a1 = (A1); e_.a2 = (A2); e_.g = (G);                                \
                if constexpr (!WIDE) e_.pad = 0;                                                              \
                cont[cs_] = e_;                                                                               \
            }                                                                                                 \
            ccount += __popcll(cm_);                                                                          \
        }                                                                                                     \
    }
    for (int phase = 0; phase < 2; ++phase) {
        Top2 top = {-INFINITY, -INFINITY, -1};
        unsigned flag = 0;
        int qcount = 0;
        // rows to sweep: bit r of `live` (row r covers boundaries 63 r .. 63 r + 63, i.e. the stretches lo .. hi of the
        // lanes' samples; it is skipped when all of them are dead at this phase's pruning level).  Lane r works that out
        // for row r, a ballot makes the mask.  Windows that are not hit-like sweep rows 0 .. rows-1.
        unsigned long long live = ~0ull;
        if (hitlike) {
            const unsigned long long dead = __ballot(cbound < Tprune);
            const float rr_ = 1.0f / static_cast<float>(rows);
            const int lo = static_cast<int>((static_cast<float>(BS_STRIDE * lane) + 0.5f) * rr_);          // exact for these small integers
            const int hi = min(63, static_cast<int>((static_cast<float>(BS_STRIDE * lane + BS_STRIDE) + 0.5f) * rr_));
            const unsigned long long span = (hi - lo >= 63) ? ~0ull : (((1ull << (hi - lo + 1)) - 1ull) << lo);
            live = __ballot(lane < rows && (dead & span) != span);
        }
        auto take_row = [&]() {                        // next live row, -1: none left (uniform)
            if (live == 0ull) return -1;
            const int r = __builtin_ctzll(live);
            live &= live - 1ull;
            return r;
        };
        auto drain = [&]() {
            // drain: interior candidates of the queued blocks
            ps_sync<64>();
            PS_STAMP_AT(wk, 1);                    // boundary sweep
            for (int r = 0; r < qcount; r += 64) {
                // (a) one queued block per lane: its 8 samples, as int16 offsets from m, go to LDS
                const int nb = min(64, qcount - r);
                if (lane < nb) {
                    const int64_t gq = base + bs_q_j(queue[r + lane], ps) - 8;
                    int y[8];
#pragma unroll
                    for (int w = 0; w < 8; ++w) y[w] = load_count<DT>(c, gq + w, bad) - m;
                    if constexpr (WIDE) {
                        ybuf[2 * lane] = make_int4(y[0], y[1], y[2], y[3]);
                        ybuf[2 * lane + 1] = make_int4(y[4], y[5], y[6], y[7]);
                    } else {
                        int4 pk;
                        pk.x = (y[0] & 0xffff) | (y[1] << 16); pk.y = (y[2] & 0xffff) | (y[3] << 16);
                        pk.z = (y[4] & 0xffff) | (y[5] << 16); pk.w = (y[6] & 0xffff) | (y[7] << 16);
                        ybuf[lane] = pk;
                    }
                }
                ps_sync<64>();
                // (b) one (block, offset) pair per lane: candidate J - u, u = 1..7 (the block's last u samples removed)
                for (int r0 = 0; r0 < nb * 7; r0 += 64) {
                    const int idx = r0 + lane;
                    const bool valid = idx < nb * 7;
                    const int eidx = valid ? idx / 7 : 0, u = idx - (idx / 7) * 7 + 1;
                    const BsQ_t q = queue[r + eidx];
                    const int J = bs_q_j(q, ps) - u;
                    s1_t x1 = bs_q_a1(q);
                    s2_t x2;
                    if constexpr (WIDE) {
                        const int4 pa = ybuf[2 * eidx], pb = ybuf[2 * eidx + 1];
                        const int yy[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
                        long long sq = 0;                       // 7 * 2^46
#pragma unroll
                        for (int w = 1; w < 8; ++w)
                            if (8 - w <= u) { x1 -= yy[w]; sq += static_cast<long long>(yy[w]) * static_cast<long long>(yy[w]); }
                        x2 = q.a2 - sq;
                    } else {
                        const int4 pk = ybuf[eidx];
                        const int w4[4] = {pk.x, pk.y, pk.z, pk.w};
                        unsigned sq = 0;                        // 7 * BS_WIDE^2 < 2^32
#pragma unroll
                        for (int w = 1; w < 8; ++w) {
                            const int y = (w & 1) ? (w4[w >> 1] >> 16) : static_cast<int>(static_cast<short>(w4[w >> 1] & 0xffff));
                            if (8 - w <= u) { x1 -= y; sq += static_cast<unsigned>(y * y); }
                        }
                        x2 = q.a2 - static_cast<double>(sq);
                    }
                    const BsEval o = bs_eval(bs_d(x1), bs_d(x2), bs_d(static_cast<s1_t>(T1 - x1)), bs_d(static_cast<s2_t>(T2 - x2)),
                                             J - ps, pe - J, cc, vfloor);                             // 1 <= J - ps < n here
                    const bool inr = valid && static_cast<unsigned>(J - cand_lo) <= crange;
                    const bool ok = o.okL && o.okR;
                    const float gq = (inr && ok) ? o.g : -INFINITY;
                    top2_push(top, gq, J - ps);
                    flag |= static_cast<unsigned>(inr && !ok);
                    if (phase) PS_COLLECT(gq >= Tc, gq, J, x1, x2)
                }
                ps_sync<64>();
            }
            qcount = 0;
            PS_STAMP_AT(wk, 2);                        // drain
        };
        // row r: lane L takes boundary t = 63 r + L (J = g0 + 8 t)
        auto do_row = [&](int r, const int4 &cur, const BsOff_t &off) {
            const bool first_row = r == 0;
            const int J = g0 + 8 * (BS_STRIDE * r + lane), nl = J - ps;
            const float nlf = static_cast<float>(nl);
            const double nld = static_cast<double>(nl);
            const s1_t a1 = bs_s1<WIDE>(cur) + off.o1;
            const s2_t a2 = bs_s2<WIDE>(cur) + off.o2;
            // screened gain of the boundary (bs_eval, with the running nl and the right side from the totals)
            const double a1d = bs_d(a1), b1d = T1d - a1d;                  // (exact: |S1| < 2^53)
            double a2d, b2d;
            if constexpr (WIDE) { a2d = d_of_i64(a2); b2d = d_of_i64(T2 - a2); }
            else { a2d = a2; b2d = T2 - a2; }
            const double DL = fma(nld, a2d, -(a1d * a1d));
            const double DR = fma(dn - nld, b2d, -(b1d * b1d));
            const float nrf = nf - nlf;
            const f2 D = {static_cast<float>(DL), static_cast<float>(DR)};
            const f2 nv = {nlf, nrf};
            const f2 rr = {__builtin_amdgcn_rcpf(nlf), __builtin_amdgcn_rcpf(nrf)};
            const f2 u = D * rr * rr;
            const f2 lgu = {__builtin_amdgcn_logf(u.x), __builtin_amdgcn_logf(u.y)};
            const f2 lg = lgu - cc;
            const f2 tt = nv * lg;
            const float g = -(tt.x + tt.y);
            const bool valid = static_cast<unsigned>(nl - 1) < static_cast<unsigned>(n - 1);    // 1 <= nl <= n-1 (false past the end)
            const bool okL = valid && u.x >= vfloor, okR = valid && u.y >= vfloor;
            // the boundary itself as a candidate (lane 0 of rows > 0 repeats a boundary already counted)
            const bool inr = static_cast<unsigned>(J - cand_lo) <= crange && (lane != 0 || first_row);
            const float ge = (inr && okL && okR) ? g : -INFINITY;
            top2_push(top, ge, nl);
            flag |= static_cast<unsigned>(inr && !(okL && okR));
            // the block (J - 8, J): left side bounded from boundary t-1 (the lane below), right side from this one
            const float aL = from_lane_below(lg.x), rlb = from_lane_below(rr.x);
            const bool pokL = from_lane_below(static_cast<int>(okL)) != 0;
            const bool blk = lane >= 1 && static_cast<unsigned>(J - 1 - cand_lo) <= crange + 6u;   // has interior candidates
            const float nl0f = nlf - 8.0f, nlef = nlf - 1.0f;
            const float nr0f = nrf + 8.0f, nref = nrf + 1.0f;
            const float bR = lg.y, rre = rr.y;
            const float cR0 = bR - 8.0f * LOG2E * rre;
            const float cR1 = bR - LOG2E * rre;
            const float cL1 = aL - 7.0f * LOG2E * rlb;
            const float h0 = -fmaf(nl0f, aL, nr0f * cR0);
            const float h1 = -fmaf(nlef, cL1, nref * cR1);
            const bool pruned = pokL && okR && nl >= 9 && fmaxf(h0, h1) < Tprune;
            const bool keep = blk && !pruned;
            const unsigned long long km = __ballot(keep);
            if (km) {
                if (keep) {
                    BsQ_t q;
                    bs_q_put(q, J, ps, a1, a2);
                    queue[qcount + lanes_below(km)] = q;
                }
                qcount += __popcll(km);
            }
            if (phase) PS_COLLECT(ge >= Tc, ge, J, a1, a2)
        };
        // rows in groups of BS_G, double-buffered: the next group's loads are in flight while this one is evaluated
        int4 ga[BS_G], gb[BS_G];
        BsOff_t offs[BS_G];                            // the group's chunk offsets: one LDS round trip per group, not per row
        auto row_load = [&](int r) { return r == 0 ? row0 : bsw[min(max(r, 0) * BS_STRIDE + lane, nblk)]; };
        auto row_off = [&](int r) { return coff[min((gbl + max(r, 0) * BS_STRIDE + lane) >> 8, nch - 1)]; };
        if (!hitlike) {
            // every row, five at a time in straight-line code (rows past the end are inert): the compiler interleaves the
            // rows of a group, which is worth 20 % of a window -- no branch may stand between them
#pragma unroll
            for (int i = 0; i < BS_G; ++i) ga[i] = row_load(i);
            for (int r0 = 0; r0 < rows; r0 += 2 * BS_G) {
                if (qcount > BS_QN - 64 * BS_G) drain();
#pragma unroll
                for (int i = 0; i < BS_G; ++i) offs[i] = row_off(r0 + i);
#pragma unroll
                for (int i = 0; i < BS_G; ++i) gb[i] = row_load(r0 + BS_G + i);
#pragma unroll
                for (int i = 0; i < BS_G; ++i) do_row(r0 + i, ga[i], offs[i]);
                if (r0 + BS_G >= rows) break;
                if (qcount > BS_QN - 64 * BS_G) drain();
#pragma unroll
                for (int i = 0; i < BS_G; ++i) offs[i] = row_off(r0 + BS_G + i);
#pragma unroll
                for (int i = 0; i < BS_G; ++i) ga[i] = row_load(r0 + 2 * BS_G + i);
#pragma unroll
                for (int i = 0; i < BS_G; ++i) do_row(r0 + BS_G + i, gb[i], offs[i]);
            }
        } else {
            // live rows only (a window that holds a split: typically 2 .. 4 of 20); an empty slot of a group loads row 0
            // again, so that no branch stands between the loads
            int ra[BS_G], rb[BS_G];                    // (uniform) row indices of the two groups, -1: none
#pragma unroll
            for (int i = 0; i < BS_G; ++i) ra[i] = take_row();
#pragma unroll
            for (int i = 0; i < BS_G; ++i) ga[i] = row_load(ra[i]);
            for (;;) {
                if (ra[0] < 0) break;
                if (qcount > BS_QN - 64 * BS_G) drain();
#pragma unroll
                for (int i = 0; i < BS_G; ++i) offs[i] = row_off(ra[i]);
#pragma unroll
                for (int i = 0; i < BS_G; ++i) rb[i] = take_row();
#pragma unroll
                for (int i = 0; i < BS_G; ++i) gb[i] = row_load(rb[i]);
#pragma unroll
                for (int i = 0; i < BS_G; ++i) if (ra[i] >= 0) do_row(ra[i], ga[i], offs[i]);
                if (rb[0] < 0) break;
                if (qcount > BS_QN - 64 * BS_G) drain();
#pragma unroll
                for (int i = 0; i < BS_G; ++i) offs[i] = row_off(rb[i]);
#pragma unroll
                for (int i = 0; i < BS_G; ++i) ra[i] = take_row();
#pragma unroll
                for (int i = 0; i < BS_G; ++i) ga[i] = row_load(ra[i]);
#pragma unroll
                for (int i = 0; i < BS_G; ++i) if (rb[i] >= 0) do_row(rb[i], gb[i], offs[i]);
            }
        }
        drain();
        if (phase == 1) break;
        // wave top-2 (DPP) and flags (ballot)
#define PS_STEP(CTRL, RM) { const float ob = dpp_movf<CTRL, RM>(-INFINITY, top.b), os = dpp_movf<CTRL, RM>(-INFINITY, top.s); \
                            const int oi = dpp_mov<CTRL, RM>(-1, top.i); top2_merge(top, ob, os, oi); }
        PS_DPP_STEPS(PS_STEP)
#undef PS_STEP
        const float ab = __shfl(top.b, 63), as = __shfl(top.s, 63);
        const int ai = __shfl(top.i, 63);
        anyflag = __ballot(flag != 0) != 0ull;
        if (!anyflag) {
            if (ab < thr_log2 - dthr) result = -1;
            else if (ab > thr_log2 + dthr && as < ab - 2.0f * dlt) result = ps + ai;
        }
#ifdef PS_NOEXACT
        if (result == -2) result = ab < thr_log2 ? -1 : ps + ai;       // timing experiment only: never the product
#endif
        PS_STAMP_AT(wk, 3);                            // top-2 reduce + decision
        if (result != -2 || anyflag) break;
        // Ambiguous for the screen (inside the threshold band, or a near tie): collect the contenders.
        // The screen is within delta of the reference gain, so the reference's choice has a screened
        // gain of at least Tc; blocks whose bound is below Tc - 2 delta hold none of them.
        Tc = fmaxf(thr_log2 - dthr, ab - 2.0f * dlt) - dlt;
        Tprune = Tc - 2.0f * dlt;
    }
#undef PS_COLLECT
    // The contenders are decided from UNCENTRED sums (sum k, sum k^2 as the reference forms them): exact only while
    // n * max|k|^2 < 2^53.  Beyond that (a large DC offset on a fine grid) the window takes the whole-window fp64 scan.
    // Wide digest: the data are a re-quantised float64 current, which the reference's own fp64 sums do not represent
    // exactly either; the contenders are decided in the same fp64 formulas from the exact 64-bit sums ABOUT m (each
    // rounded to fp64 once: closer to the true value than any order of summation).
    const bool sums_exact = WIDE || static_cast<double>(n) * static_cast<double>(mabs) * static_cast<double>(mabs) < 9007199254740992.0;
    if (result == -2 && !anyflag && ccount <= BS_NC && sums_exact) {
        ps_sync<64>();                              // contender stores visible to the other lanes
        double var_summed;
        if constexpr (WIDE) var_summed = static_cast<double>(n) * log(ref_var(T1d, T2d, n, c.q, c.q2));
        else var_summed = static_cast<double>(n) *
            log(ref_var(T1d + dn * static_cast<double>(m),
                        T2d + 2.0 * static_cast<double>(m) * T1d + dn * static_cast<double>(m) * static_cast<double>(m), n, c.q, c.q2));
        double eg = thresh;
        int ei = -1;
        if (lane < ccount) {
            const BsC_t e = cont[lane];
            double gx;
            if constexpr (WIDE) {
                const int nl = e.j - ps;
                gx = ref_gain(var_summed, nl, ref_var(d_of_i64(e.a1), d_of_i64(e.a2), nl, c.q, c.q2),
                              n - nl, ref_var(d_of_i64(T1 - e.a1), d_of_i64(T2 - e.a2), n - nl, c.q, c.q2));
            } else {
                gx = bs_exact_gain(c, m, e.a1, e.a2, T1, T2, e.j - ps, n, var_summed);
            }
            if (gx > eg) { eg = gx; ei = e.j; }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const double og = __shfl_xor(eg, d);
            const int oi = __shfl_xor(ei, d);
            if (beats(og, oi, eg, ei)) { eg = og; ei = oi; }
        }
        result = ei;
        if (lane == 0) wk.exact += 1;
    }
    PS_STAMP_AT(wk, 4);                                // contenders + fp64 decision
#ifdef PS_STAMP
    if (lane == 0 && anyflag) wk.ph[10] += 1;
    if (lane == 0 && !anyflag && result == -2) wk.ph[11] += 1;
#endif
    if (c.mode == MODE_VERIFY || result == -2) {
        if (lane == 0) wk.exact += (1LL << 32);                        // high word: full exact scans
        const int ex = scan_exact<64, DT>(c, nullptr, base + ps, ps, n, cand_lo, cand_hi, thresh, nullptr, sh, bad, nullptr);
        if (result != -2 && ex != result) {
            bad |= ST_VERIFY_MISMATCH;
            if (lane == 0 && wk.dbg[1] == 0) { wk.dbg[0] = ps; wk.dbg[1] = pe; wk.dbg[2] = result; wk.dbg[3] = ex; }
        }
        PS_STAMP_AT(wk, 6);
        return ex;
    }
    return result;
}

// ---- K2 from the K0 digest: per-segment statistics without a second pass over the samples ------------------
// Segment.mean/std/min/max (core.py:209-223).  One wave per segment (workgroups stride over the segments):
// S1, S2 of the full blocks inside the segment from the chunk prefix and the chunk totals, min/max from the per-block
// table, the ragged ends (<= 7 samples each) from the samples.  mean = (m + S1/n) q, std = sqrt(S2/n - (S1/n)^2) q
// (population; formed about m, so nothing cancels), min/max exact.  n_seg = bounds_off[n_ev] + n_ev is read on the
// device; `stats_cap` bounds the writes.
template <int DT>
__global__ __launch_bounds__(64) void segstat_bs_kernel(DevCfg c, const int64_t *ev_start, const int64_t *ev_len, int32_t n_ev,
                                                        const int32_t *bounds, const int64_t *bounds_off, ps_segstat *stats,
                                                        int64_t stats_cap, unsigned *status, const AsmHeader *hdr)
{
    const int lane = threadIdx.x;
    // a failed stitch or a refused digest leaves no valid boundaries: the host redoes the call on another path
    if ((hdr && hdr->fail) || (*status & ~ST_VERIFY_MISMATCH) != 0u) return;
    const int64_t n_seg = min(bounds_off[n_ev] + n_ev, stats_cap);
    unsigned bad = 0;
    for (int64_t g = blockIdx.x; g < n_seg; g += gridDim.x) {
        int lo = 0, hi = n_ev - 1;                 // event e: bounds_off[e] + e <= g < bounds_off[e+1] + e + 1
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (bounds_off[mid] + mid <= g) lo = mid; else hi = mid - 1;
        }
        const int e = lo;
        const int64_t boff = bounds_off[e];
        const int cnt = static_cast<int>(bounds_off[e + 1] - boff);
        const int sidx = static_cast<int>(g - boff - e);
        const int a = sidx == 0 ? 0 : bounds[boff + sidx - 1];
        const int b = sidx == cnt ? static_cast<int>(ev_len[e]) : bounds[boff + sidx];
        const int64_t base = ev_start[e];
        const int4 info = c.ev_info[e];
        const int m = info.x;
        const long long eb = (static_cast<long long>(static_cast<unsigned>(info.w)) << 32) | static_cast<unsigned>(info.z);
        const int n = b - a;
        if (a < 0 || b < a || b > ev_len[e]) continue;                     // (never with valid boundaries)
        double s1 = 0.0, s2 = 0.0;                 // sums of y = k - m and y^2 (exact integers)
        int mn = 0x7fffffff, mx = static_cast<int>(0x80000000);
        const int b0 = (a + 7) >> 3, b1 = b >> 3;  // full blocks [b0, b1) of the event
        if (n < 32 || b0 >= b1) {
            for (int i = a + lane; i < b; i += 64) {
                const int y = load_count<DT>(c, base + i, bad) - m;
                s1 += static_cast<double>(y); s2 += static_cast<double>(y) * static_cast<double>(y);
                mn = min(mn, y); mx = max(mx, y);
            }
        } else {
            const long long gb0 = eb + b0, gb1 = eb + b1;
            // ragged ends from the samples: lanes 0..6 the head [a, 8 b0), lanes 32..38 the tail [8 b1, b)
            int y = 0;
            bool have = false;
            if (lane < 8 * b0 - a) { y = load_count<DT>(c, base + a + lane, bad) - m; have = true; }
            if (lane >= 32 && lane - 32 < b - 8 * b1) { y = load_count<DT>(c, base + 8 * b1 + (lane - 32), bad) - m; have = true; }
            if (have) { s1 = static_cast<double>(y); s2 = static_cast<double>(y) * static_cast<double>(y); mn = y; mx = y; }
            // chunk totals between the two boundaries, min/max of the full blocks
            const long long c0 = gb0 >> 8, c1 = gb1 >> 8;
            for (long long cc = c0 + lane; cc < c1; cc += 64) {
                const int4 t = c.chunk_tot[2 * cc];
                s1 += static_cast<double>(t.x); s2 += ent2(t);
            }
            for (long long gb = gb0 + lane; gb < gb1; gb += 64) {
                const int w = c.blk_mm[gb];
                mn = min(mn, static_cast<int>(static_cast<short>(w & 0xffff))); mx = max(mx, w >> 16);
            }
            if (lane == 0) {
                const int4 p0 = c.bsum[gb0], p1 = c.bsum[gb1];
                s1 += static_cast<double>(p1.x) - static_cast<double>(p0.x);
                s2 += ent2(p1) - ent2(p0);
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            s1 += __shfl_down(s1, d); s2 += __shfl_down(s2, d);
            mn = min(mn, __shfl_down(mn, d)); mx = max(mx, __shfl_down(mx, d));
        }
        if (lane == 0) {
            ps_segstat r;
            if (n > 0) {
                const double dn = static_cast<double>(n);
                const double my = s1 / dn;
                double var = s2 / dn - my * my;
                if (var < 0) var = 0;
                r.mean = (static_cast<double>(m) + my) * c.q; r.std = sqrt(var) * c.q;
                r.min = static_cast<double>(m + mn) * c.q; r.max = static_cast<double>(m + mx) * c.q;
            } else {
                r.mean = r.std = r.min = r.max = __builtin_nan("");
            }
            stats[g] = r;
        }
    }
    if (bad) atomicOr(status, bad);
}

}  // namespace ps
