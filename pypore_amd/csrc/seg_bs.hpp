// seg_bs.hpp -- block-sum window scan: one WAVE scans one window, no LDS image of the samples.
//
// K0 (blocksum_kernel) streams the trace once.  Per 8 samples it forms S1 = sum (k-m) and S2 = sum (k-m)^2
// (m = first count of the event) and leaves their EXCLUSIVE prefix within each chunk of 128 blocks -- 8 bytes per
// block: 25 + 39 bits (|k-m| < 2^14), or 16 bytes (two int64) for the wide digest -- plus the chunk totals.  The exact
// sums of [ps, J) at any block boundary J of a window are then one coalesced load plus a per-chunk offset that the
// window keeps in registers: a window scan needs no per-sample work for the blocks the bound discards (98 % of them).
// Lane L of the wave takes boundaries L, L+63, ...: the boundary candidates are evaluated from the sums
// (D = n*S2 - S1^2 in fp64, everything after it in fp32), the monotone block bound prunes with the neighbouring lane's
// values, and only surviving blocks read raw samples.  No barrier, no cross-wave reduction.
//
// Round 3: 8-byte digest (was 16), chunks of 128 blocks handled by half a wave in K0 (four consecutive
// blocks per thread after a transposition through LDS: one half-wave scan per four blocks instead of a workgroup scan
// per block), chunk offsets of a window in registers (v_readlane per row instead of an LDS table), the window's event
// constants passed in by the caller (no per-window load of the event table), a scan body written for 128 registers
// (four waves per SIMD instead of two) with its cold paths out of line.
//
// Round 4: one record per group of 32 blocks from K0 (the exact sums at the group's first boundary and
// two amplitudes of its prefix path about its chord); every window scan of the narrow digest starts with a COARSE PASS over
// those records -- the gain is convex in (k, S1, S2), so a whole group is bounded from its two boundaries -- and sweeps only
// the rows that hold a group it could not discard (3 of a window's 16); the bounds are audited on the device (AUDIT
// instance, audit_kernel).  A ring of two digest rows instead of four.
//
// Reference functions restated here: cparsers.pyx:157-178 (_best_split_stepwise) through scan_window_bs,
// core.py:209-223 (Segment statistics) through segstat_bs_kernel.
//
// Included by seg_device.hpp after the common helpers (screen arithmetic, DPP primitives, scan_exact).
#pragma once
#include <type_traits>

namespace ps {

constexpr int BS_WIDE = 16384;             // narrow digest: |k - m| must stay below this (a block's S2 < 2^31, a chunk prefix:
                                           // |S1| < 2^24 -> 25 bits, S2 < 2^38 -> 39 bits; a window of 64 chunks: |S1| < 2^31)
enum : unsigned { ST_WIDE_RANGE = 16u };
constexpr int BS_CHUNK = 128;              // blocks per prefix chunk (1024 samples): half a wave of K0
constexpr int BS_CHUNK_LOG = 7;
constexpr int BS_MAXCH = 64;               // chunks a window may touch (one lane each): windows up to 64 512 samples
// Wide digest (DT & DT_WIDE): both moments as 64-bit integers -- (E1 lo, E1 hi, E2 lo, E2 hi) per block, the chunk
// totals alike.  |k - m| < BSW_LIM: a block's S2 < 2^49, a chunk prefix < 2^56, a window of 64 512 samples < 2^62.
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
// digest entry of the narrow format: hi = (E1 << 7) | (E2 >> 32), lo = E2 & 0xffffffff
__device__ __forceinline__ int bs8_s1(uint2 v) { return static_cast<int>(v.y) >> 7; }
__device__ __forceinline__ unsigned long long bs8_s2(uint2 v) { return u64_of(v.x, v.y & 0x7fu); }
// 2^52 + E2 as a double, by bit pattern (E2 < 2^39): the caller adds (offset - 2^52), which is exact
__device__ __forceinline__ double bs8_s2_biased(uint2 v) { return __hiloint2double(static_cast<int>((v.y & 0x7fu) | 0x43300000u), static_cast<int>(v.x)); }
constexpr double BS_BIAS = 4503599627370496.0;        // 2^52
__device__ __forceinline__ uint2 bs8_pack(int e1, unsigned long long e2)
{
    return make_uint2(static_cast<unsigned>(e2), (static_cast<unsigned>(e1) << 7) | static_cast<unsigned>(e2 >> 32));
}

// ---- wave primitives of this file ---------------------------------------------------------------------
// half-wave (32 lanes) inclusive scans: row_shr 1, 2, 4, 8 inside the rows of 16, then row_bcast15 into rows 1 and 3
#define PS_DPP_HALF_STEPS(X) X(0x111, 0xf) X(0x112, 0xf) X(0x114, 0xf) X(0x118, 0xf) X(0x142, 0xa)
__device__ __forceinline__ int half_incl_scan_i32(int x)
{
#define PS_STEP(CTRL, RM) { x += dpp_mov<CTRL, RM>(0, x); }
    PS_DPP_HALF_STEPS(PS_STEP)
#undef PS_STEP
    return x;
}
__device__ __forceinline__ long long half_incl_scan_i64(long long x)
{
#define PS_STEP(CTRL, RM) { const int lo_ = dpp_mov<CTRL, RM>(0, static_cast<int>(x)), hi_ = dpp_mov<CTRL, RM>(0, static_cast<int>(x >> 32)); \
                            x += i64_of(lo_, hi_); }
    PS_DPP_HALF_STEPS(PS_STEP)
#undef PS_STEP
    return x;
}
__device__ __forceinline__ int half_max_i32(int x)            // result in lanes 31 and 63
{
#define PS_STEP(CTRL, RM) { x = max(x, dpp_mov<CTRL, RM>(static_cast<int>(0x80000000), x)); }
    PS_DPP_HALF_STEPS(PS_STEP)
#undef PS_STEP
    return x;
}
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
// Maximum over the wave of a float that is not NaN (-inf allowed), returned uniform -- through the order-preserving integer
// image of the float: the compiler folds an integer DPP move into v_max_i32_dpp, while a float maximum costs a move, a
// canonicalisation and the maximum per step (24 instructions for the six steps instead of 6).
__device__ __forceinline__ float wave_max_f32(float x)
{
    int k = __float_as_int(x);
    k ^= (k >> 31) & 0x7fffffff;
    k = __builtin_amdgcn_readlane(wave_max_i32(k), 63);
    k ^= (k >> 31) & 0x7fffffff;
    return __int_as_float(k);
}
// uniform values into scalar registers (the compiler cannot prove uniformity of what comes out of LDS or a lane read)
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ float uni(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
__device__ __forceinline__ double uni(double x)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
__device__ __forceinline__ long long uni(long long x)
{
    return i64_of(__builtin_amdgcn_readfirstlane(static_cast<int>(x)), __builtin_amdgcn_readfirstlane(static_cast<int>(x >> 32)));
}
// value of lane `l` (uniform index)
__device__ __forceinline__ int lane_get(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ double lane_get(double x, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ __forceinline__ long long lane_get(long long x, int l)
{
    return i64_of(__builtin_amdgcn_readlane(static_cast<int>(x), l), __builtin_amdgcn_readlane(static_cast<int>(x >> 32), l));
}

// Arguments and results travel by value: the address of a caller's local (its `bad` word, its counters, the kernel's
// DevCfg) would move that object to the stack for the whole kernel.  Results: low word = split, high word = status bits /
// near-tie flag.
struct BsCold { const void *samples; float inv_q; int off_counts; double q, q2, dc_counts; float noise_k; };
__device__ __forceinline__ BsCold bs_cold(const DevCfg &c) { BsCold k = {c.samples, c.inv_q, c.off_counts, c.q, c.q2, c.dc_counts, c.noise_k}; return k; }
__device__ __forceinline__ DevCfg bs_cold_cfg(const BsCold &k)
{
    DevCfg c = {};
    c.samples = k.samples; c.inv_q = k.inv_q; c.off_counts = k.off_counts; c.q = k.q; c.q2 = k.q2; c.dc_counts = k.dc_counts; c.noise_k = k.noise_k;
    return c;
}

// ---- groups (round 4): the coarse level of the window scan -----------------------------------------------------
// A group is 32 consecutive blocks (256 samples, aligned in the global block index): the eight lanes of K0 that hold four
// consecutive blocks each.  K0 leaves one 16-byte record per group: the digest entry of the group's first block (the
// exact sums at the group's left boundary) and two amplitudes (D1, D2) as fp32, rounded up:
//     D1 >= max_j | sum_{i<j} (y_i - mu_g) |,     D2 >= max_j | sum_{i<j} ((y_i - mu_g)^2 - v_g) |       (0 <= j <= 256)
// -- how far the prefix sums of the group's centred samples and of their squares stray from their chords (mu_g, v_g:
// the group's mean and variance).  Both are formed at the 31 interior block boundaries from the block sums; inside block b
// the bridges leave the chord between its two boundaries by at most sqrt(2 q_b) and 7/8 q_b, q_b = sum over the block of
// (y - mu_g)^2 (blocksum_kernel has the derivation), so
//     D1 = max_b (max(|d1(8b)|, |d1(8b+8)|) + sqrt(2 q_b)),   D2 = max_b (max(|d2z(8b)|, |d2z(8b+8)|) + 7/8 q_b).
// The gain of a candidate is a convex function of the left part's (k, S1, S2) (seg_bs.hpp: bs_group_slack), so these two
// numbers bound the gain of all 255 candidates inside the group from the evaluations of its two boundaries.
constexpr int BS_GRP_LOG = 5;                            // blocks per group: 32
constexpr int BS_GRP = 1 << BS_GRP_LOG;
constexpr int BS_GRP_SAMPLES = 8 * BS_GRP;
// sum / max / min over the eight lanes 8 j .. 8 j + 7, in all of them (quad_perm xor 1, xor 2, row_half_mirror)
#define PS_OCT_STEPS(X) X(0xB1) X(0x4E) X(0x141)
__device__ __forceinline__ int oct_allsum(int x)
{
#define PS_STEP(CTRL) { x += dpp_mov<CTRL, 0xf>(0, x); }
    PS_OCT_STEPS(PS_STEP)
#undef PS_STEP
    return x;
}
// (the `old` operand is the operation's identity, as in the wave scans: the compiler then folds the move into the
//  operation -- v_max_i32_dpp -- instead of copying the register first; every lane has a source in these three controls)
__device__ __forceinline__ int oct_allmax(int x)
{
#define PS_STEP(CTRL) { x = max(x, dpp_mov<CTRL, 0xf>(static_cast<int>(0x80000000), x)); }
    PS_OCT_STEPS(PS_STEP)
#undef PS_STEP
    return x;
}
__device__ __forceinline__ int oct_allmin(int x)
{
#define PS_STEP(CTRL) { x = min(x, dpp_mov<CTRL, 0xf>(0x7fffffff, x)); }
    PS_OCT_STEPS(PS_STEP)
#undef PS_STEP
    return x;
}
__device__ __forceinline__ float oct_allmaxf(float x)                       // x >= 0
{
#define PS_STEP(CTRL) { x = __uint_as_float(max(__float_as_uint(x), static_cast<unsigned>(dpp_mov<CTRL, 0xf>(0, static_cast<int>(__float_as_uint(x)))))); }
    PS_OCT_STEPS(PS_STEP)
#undef PS_STEP
    return x;
}
__device__ __forceinline__ float pmaxf(float a, float b) { return __int_as_float(max(__float_as_int(a), __float_as_int(b))); }   // a, b >= 0, no NaN
// value of the first lane of this lane's group of eight (quad broadcast, then row_shr:4 for the upper quad)
__device__ __forceinline__ int oct_first(int x, int lane)
{
    const int q = __builtin_amdgcn_update_dpp(0, x, 0x00, 0xf, 0xf, true);
    const int s = __builtin_amdgcn_update_dpp(0, q, 0x114, 0xf, 0xf, true);
    return (lane & 4) ? s : q;
}

// ---- K0 -----------------------------------------------------------------------------------------------
// Event e owns the blocks [ev_boff[e], ev_boff[e+1]) of the global block index gb (events back to back in block units;
// a last partial block is padded with k = m).  A wave takes 256 consecutive blocks = two chunks:
//   1. lane L loads blocks L, 64+L, 128+L, 192+L (coalesced: 2 KB per load instruction and block row) and forms their
//      S1, S2, min, max -- per-sample work kept to what the digest needs: fp32 samples are checked for integrality in
//      float (x/q, v_rndne_f32, the difference OR-ed into one word), counts pass through min3 / max3 / add3 and one
//      24-bit multiply-add each; the range check (|k - m| < BS_WIDE) once per block from its min and max;
//   2. the four records go through LDS (wave-local, no barrier) so that lane L then holds blocks 4L .. 4L+3: their
//      prefix is three additions, and ONE half-wave scan (lanes 0..31 = chunk A, 32..63 = chunk B) per four blocks
//      completes the chunk-exclusive prefix -- the second moments as two 16-bit halves in fused integer DPP additions;
//   3. lane L writes its four consecutive digest entries (32 bytes) and, when statistics are wanted, the four min/max
//      words; lanes 31 and 63 write the chunk totals (S1, max |k-m|, S2 as int64).
// bs[gb]: prefix of the blocks of gb's chunk that precede gb (entries up to one past the last block are valid: the end
// boundary of the last window; the arrays are padded to whole waves).  A chunk may straddle events (different m): only
// differences inside one event are ever formed.
#ifndef PS_K0_AMP
#define PS_K0_AMP 2                                    // how K0 forms the group amplitudes: 2 (default) block sums of squares about the group mean, fp32 bridges at the
#endif                                                 // block boundaries, slack inside a block from its own sum of squares; 1 (first form of round 4) bridges in fp64,
                                                       // slack from the group's min / max -- looser (3.8 instead of 3.6 rows of a window without a split), 20 more fp64
                                                       // instructions per lane: 0.2262 -> 0.2227 ms per step in five interleaved rounds (see blocksum_kernel)
#ifndef PS_K0_TRIM
#define PS_K0_TRIM 1                                   // instruction trims of the streaming route (second half of round 4): sums of squares as a multiply-add chain, one
#endif                                                 // range check per lane instead of one per block, integer maxima of non-negative floats; 0 builds the code before them
#ifndef PS_K0_MINW
#define PS_K0_MINW 4                                   // waves per SIMD K0 is compiled for: two sets of eight 16-byte loads per lane (round 5: persistent, one or two waves per SIMD are launched)
#endif
constexpr int K0_BPT = 4;                              // consecutive blocks per thread after the transposition
constexpr int K0_WB = 64 * K0_BPT;                     // blocks per wave
#ifndef PS_K0_WAVES
#define PS_K0_WAVES 4
#endif
#ifndef PS_K0_PKF32
#define PS_K0_PKF32 1                                  // fp32 samples on the narrow digest: packed conversion + the packed int16 block sums (k0_block_sums_pkf32); 0: round 4's form
#endif
#ifndef PS_K0_PK16
#define PS_K0_PK16 1                                   // int16 samples on the narrow digest: block sums on packed int16 arithmetic (k0_block_sums_pk16); 0: unpacked, as in round 4
#endif
#ifndef PS_K0_PRIO
#define PS_K0_PRIO 0                                   // s_setprio of K0's waves (0: none)
#endif
constexpr int K0_WAVES = PS_K0_WAVES;                  // waves per workgroup (independent of each other)
__host__ __device__ inline long long k0_padded_blocks(long long nb_total) { return (nb_total + 1 + K0_WB - 1) / K0_WB * K0_WB; }

// Wave-local hand-over through LDS: the lanes of a wave run in lockstep and its LDS operations complete in order, so the
// exchange only needs the writes to be complete (lgkmcnt) -- NOT the vector-memory counter: a workgroup-scope fence would also
// wait for the samples of the next wave block, which are in flight on purpose.
__device__ __forceinline__ void k0_lds_ready() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ void k0_lds_done() { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }

template <int DT> struct K0Rec { int s1; unsigned s2; unsigned long long s2w; int ymin, ymax; };
template <int DT> struct K0Gen { K0Rec<DT> r; int m; unsigned bad; };     // what the general route returns (by value: no stack objects)

// sums of one block from its eight offsets y = k - m
template <int DT>
__device__ __forceinline__ void k0_block_sums(const int (&y)[8], K0Rec<DT> &r)
{
    constexpr bool WIDE = bs_wide<DT>();
    r.s1 = 0; r.s2 = 0; r.s2w = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        r.s1 += y[q];
        if constexpr (WIDE) r.s2w += static_cast<unsigned long long>(static_cast<long long>(y[q]) * static_cast<long long>(y[q]));
#if !PS_K0_TRIM
        else r.s2 += static_cast<unsigned>(__mul24(y[q], y[q]));           // |y| < BS_WIDE < 2^23, else the call is redone
#endif
    }
#if PS_K0_TRIM
    if constexpr (!WIDE) {
        // one multiply and seven multiply-adds (left to itself the compiler forms seven products and sums them in threes: 11)
        unsigned acc = static_cast<unsigned>(__mul24(y[0], y[0]));         // |y| < BS_WIDE < 2^23, else the call is redone
#pragma unroll
        for (int q = 1; q < 8; ++q) asm("v_mad_i32_i24 %0, %1, %1, %2" : "=v"(acc) : "v"(y[q]), "v"(acc));
        r.s2 = acc;
    }
#endif
    r.ymin = min(min(min(y[0], y[1]), min(y[2], y[3])), min(min(y[4], y[5]), min(y[6], y[7])));
    r.ymax = max(max(max(y[0], y[1]), max(y[2], y[3])), max(max(y[4], y[5]), max(y[6], y[7])));
}

// eight aligned samples (raw: 32 bytes fp32 / 16 bytes int16) -> offsets from m
template <int DT>
__device__ __forceinline__ void k0_offsets(const DevCfg &c, const int4 *raw, int m, float mf, unsigned &nz, int (&y)[8])
{
    if (sdt(DT) == PS_DTYPE_F32) {
        const f2 iq = {c.inv_q, c.inv_q};
        const f2 mf2 = {mf, mf};
#pragma unroll
        for (int v = 0; v < 2; ++v) {
#pragma clang fp contract(off)                                              // (x/q rounded first, as in to_count)
            const int w[4] = {raw[v].x, raw[v].y, raw[v].z, raw[v].w};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f2 x = {__int_as_float(w[2 * h]), __int_as_float(w[2 * h + 1])};
                const f2 t = x * iq;
                const f2 r = {__builtin_rintf(t.x), __builtin_rintf(t.y)};
                const f2 d = t - r;                                     // +0 exactly when t is an integer; NaN for inf/NaN
                nz |= __float_as_uint(d.x) | __float_as_uint(d.y);
                const f2 yf = r - mf2;                                  // exact while |r| < 2^24 (else: saturates below -> wide)
                y[v * 4 + 2 * h] = static_cast<int>(yf.x);
                y[v * 4 + 2 * h + 1] = static_cast<int>(yf.y);
            }
        }
    } else {
        const int w[4] = {raw[0].x, raw[0].y, raw[0].z, raw[0].w};
        const int om = c.off_counts - m;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            y[q] = ((q & 1) ? (w[q >> 1] >> 16) : static_cast<int>(static_cast<short>(w[q >> 1] & 0xffff))) + om;
    }
}

// int16 samples, narrow digest (round 5): a block's sums on PACKED int16 arithmetic -- no per-sample unpack.  The eight counts
// of a block are four dwords of two; k - k0 (k0: the event's first raw count, so k - k0 = (k + off) - m) by v_pk_sub_i16 with
// clamp: a difference beyond int16 saturates, which the range check |y| < BS_WIDE (2^14) then sees -- the call is redone
// on the wide digest like any other such call --; S1 and S2 by v_dot2_i32_i16 (two samples per instruction, against (1, 1)
// and against itself: 8 x 16383^2 < 2^31), the extremes by v_pk_min / v_pk_max_i16.  13 instructions less per sample pair.
typedef short k0_v2s __attribute__((ext_vector_type(2)));
// a.b without an accumulator (the three-operand form with a literal zero: the builtin took a v_mov_b32 for the zero first)
__device__ __forceinline__ int k0_dot2_first(k0_v2s a, k0_v2s b)
{
    int out;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(out) : "v"(a), "v"(b));
    return out;
}

template <int DT>
__device__ __forceinline__ void k0_block_sums_pk16(const int4 raw, int k0pk, K0Rec<DT> &r)
{
    const k0_v2s k0v = __builtin_bit_cast(k0_v2s, k0pk);
    const k0_v2s ones = {1, 1};
    const int w[4] = {raw.x, raw.y, raw.z, raw.w};
    k0_v2s y[4];
    int s1 = 0, s2 = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        y[q] = __builtin_elementwise_sub_sat(__builtin_bit_cast(k0_v2s, w[q]), k0v);
        s1 = q == 0 ? k0_dot2_first(y[q], ones) : __builtin_amdgcn_sdot2(y[q], ones, s1, false);
        s2 = q == 0 ? k0_dot2_first(y[q], y[q]) : __builtin_amdgcn_sdot2(y[q], y[q], s2, false);
    }
    const k0_v2s mn = __builtin_elementwise_min(__builtin_elementwise_min(y[0], y[1]), __builtin_elementwise_min(y[2], y[3]));
    const k0_v2s mx = __builtin_elementwise_max(__builtin_elementwise_max(y[0], y[1]), __builtin_elementwise_max(y[2], y[3]));
    r.s1 = s1; r.s2 = static_cast<unsigned>(s2); r.s2w = 0;
    r.ymin = min(static_cast<int>(mn.x), static_cast<int>(mn.y));
    r.ymax = max(static_cast<int>(mx.x), static_cast<int>(mx.y));
}

// fp32 samples, narrow digest (round 5): the same packed int16 pipeline behind a packed conversion.  t = x / q is the count as a
// float (v_pk_mul_f32; integral iff v_fract_f32(t) == 0 -- NaN for NaN, and an infinity saturates below), t - m is exact
// (integers below 2^24), and v_cvt_pknorm_i16_f32((t - m) / 32767) gives round(clamp((t - m) / 32767, -1, 1) * 32767): the
// product carries 2^-24 relative, 0.002 of a count at 32767, so every |y| <= 32767 comes out exactly and everything beyond
// saturates to +-32767, which fails |y| < BS_WIDE like any wide count (tools/probes/pknorm_probe.hip checks all of
// -70 000 .. 70 000 on the device).  5.5 instructions per sample where the unpacked form took ~10.
template <int DT>
__device__ __forceinline__ void k0_block_sums_pkf32(const int4 (&raw)[2], float inv_q, float mf, unsigned &nz, K0Rec<DT> &r)
{
    const f2 iq = {inv_q, inv_q}, m2 = {mf, mf};
    const f2 cn = {1.0f / 32767.0f, 1.0f / 32767.0f};
    const k0_v2s ones = {1, 1};
    const int w[8] = {raw[0].x, raw[0].y, raw[0].z, raw[0].w, raw[1].x, raw[1].y, raw[1].z, raw[1].w};
    k0_v2s y[4];
    int s1 = 0, s2 = 0;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
#pragma clang fp contract(off)
        const f2 x = {__int_as_float(w[2 * h]), __int_as_float(w[2 * h + 1])};
        const f2 t = x * iq;
        nz |= __float_as_uint(__builtin_amdgcn_fractf(t.x)) | __float_as_uint(__builtin_amdgcn_fractf(t.y));
        const f2 a = (t - m2) * cn;
        y[h] = __builtin_amdgcn_cvt_pknorm_i16(a.x, a.y);
        s1 = h == 0 ? k0_dot2_first(y[h], ones) : __builtin_amdgcn_sdot2(y[h], ones, s1, false);
        s2 = h == 0 ? k0_dot2_first(y[h], y[h]) : __builtin_amdgcn_sdot2(y[h], y[h], s2, false);
    }
    const k0_v2s mn = __builtin_elementwise_min(__builtin_elementwise_min(y[0], y[1]), __builtin_elementwise_min(y[2], y[3]));
    const k0_v2s mx = __builtin_elementwise_max(__builtin_elementwise_max(y[0], y[1]), __builtin_elementwise_max(y[2], y[3]));
    r.s1 = s1; r.s2 = static_cast<unsigned>(s2); r.s2w = 0;
    r.ymin = min(static_cast<int>(mn.x), static_cast<int>(mn.y));
    r.ymax = max(static_cast<int>(mx.x), static_cast<int>(mx.y));
}

// One block by the general route: any event layout, partial last blocks, unaligned int16 events.  Out of line: the
// streaming route of the kernel keeps its registers to itself.
template <int DT>
__device__ __attribute__((noinline)) K0Gen<DT> k0_block_general(BsCold k, const int64_t *ev_start, const int64_t *ev_len, const int64_t *ev_boff,
                                                                int64_t n_samples, long long gb, long long nb_total, int e_hint, int4 *ev_info)
{
    const DevCfg c = bs_cold_cfg(k);
    K0Gen<DT> out;
    K0Rec<DT> &r = out.r;
    unsigned bad = 0;
    r.s1 = 0; r.s2 = 0; r.s2w = 0; r.ymin = 0; r.ymax = 0;
    out.m = 0; out.bad = 0;
    if (gb >= nb_total) return out;                    // beyond the last block: empty (the end boundary's prefix is still written)
    int e = e_hint;
    while (ev_boff[e + 1] <= gb) ++e;                  // (empty events are stepped over)
    const int64_t len = ev_len[e], base = ev_start[e];
    const long long b = gb - ev_boff[e];
    const int64_t i0 = 8 * b;
    const int m = load_count<DT>(c, base, bad);
    out.m = m;
    int y[8];
    const int cnt = static_cast<int>(len - i0 < 8 ? len - i0 : 8);
    constexpr int ES = static_cast<int>(sizeof(typename Raw<DT>::type));
    const char *p = static_cast<const char *>(c.samples) + (base + i0) * ES;
    if (cnt == 8 && (reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
        constexpr int NV = 8 * ES / 16;
        int4 raw[2];
#pragma unroll
        for (int v = 0; v < NV; ++v) raw[v] = reinterpret_cast<const int4 *>(p)[v];
        unsigned nz = 0;
        k0_offsets<DT>(c, raw, m, static_cast<float>(m), nz, y);
        if (nz) bad |= ST_OFF_GRID;
        k0_block_sums<DT>(y, r);
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
        k0_block_sums<DT>(y, r);
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) y[q] = q < cnt ? load_count<DT>(c, base + i0 + q, bad) - m : 0;   // padding = m: y = 0
        k0_block_sums<DT>(y, r);
        r.ymin = r.ymax = y[0];                    // (cnt >= 1: block b exists only if it holds a sample) over the real samples
#pragma unroll
        for (int q = 1; q < 8; ++q)
            if (q < cnt) { r.ymin = min(r.ymin, y[q]); r.ymax = max(r.ymax, y[q]); }
    }
    if (b == 0) ev_info[e] = make_int4(m, 0, static_cast<int>(ev_boff[e] & 0xffffffffLL), static_cast<int>(ev_boff[e] >> 32));
    out.bad = bad;
    return out;
}

// Round 5: the kernel is PERSISTENT.  A launch holds a fixed number of waves (host: k0_grid workgroups, one or two waves per
// SIMD) and every wave strides over the wave blocks (256 blocks = 2 048 samples each) of the call; the samples of its NEXT wave
// block are requested before the current one is worked on (a second set of load registers), so that one wave keeps 8 KB in
// flight all the time.  K0 is the only kernel that streams the samples and it is bound by HBM: a few waves per SIMD saturate
// the memory, and everything beyond that only keeps the scan waves of the other calls in flight off the SIMD (round 4: five
// 82-register waves per SIMD were 54 % of all wave residency while issuing 12 % of the time).  A launch that covers every wave
// block with a wave of its own (gridDim.x * K0_WAVES >= wave blocks) degenerates to the one-shot kernel of rounds 3 and 4.
struct K0Src { int e; int fast; long long b_first; int64_t base; };
// 16 bytes of samples from an address that is only SAMPLE-aligned (2 bytes for int16, 4 for fp32): the vector type says so, so
// that the compiler may not assume 16-byte alignment (dereferencing an int4 * there is undefined behaviour) -- with
// the target's unaligned access mode it is still ONE global_load_dwordx4 (checked in the ISA: tools/kernel_resources.py counts
// them), and ps_create probes once per device that such a load returns the right bytes (poreseg.hip: k0_unaligned_probe).
template <int A> struct K0Vec;
template <> struct K0Vec<2> { typedef int type __attribute__((ext_vector_type(4), aligned(2))); };
template <> struct K0Vec<4> { typedef int type __attribute__((ext_vector_type(4), aligned(4))); };
template <int A> __device__ __forceinline__ int4 k0_load16(const char *p)
{
    const typename K0Vec<A>::type v = *reinterpret_cast<const typename K0Vec<A>::type *>(p);
    return make_int4(v.x, v.y, v.z, v.w);
}
// the probe: 16-byte loads at every sample offset 0 .. 15 of a small pattern, against the same bytes fetched one by one
template <int A>
__global__ void k0_unaligned_probe_kernel(const unsigned char *buf, int n_off, unsigned *bad)
{
    const int o = threadIdx.x;
    if (o >= n_off) return;
    const int4 v = k0_load16<A>(reinterpret_cast<const char *>(buf) + o * A);
    const unsigned char *q = buf + o * A;
    unsigned w[4];
    for (int k = 0; k < 4; ++k) w[k] = q[4 * k] | (q[4 * k + 1] << 8) | (q[4 * k + 2] << 16) | (static_cast<unsigned>(q[4 * k + 3]) << 24);
    if (static_cast<unsigned>(v.x) != w[0] || static_cast<unsigned>(v.y) != w[1] || static_cast<unsigned>(v.z) != w[2] || static_cast<unsigned>(v.w) != w[3])
        atomicOr(bad, 1u);
}

// Round 6: NS register sets.  What K0 costs the scan kernels of the other calls in flight is the REGISTER FILE it holds while its
// bytes travel (tools/r6/residency_probe.py: waves that only hold registers slow the scans by ~0.02 ms per step per 100
// registers held per SIMD), and a wave's ~64 working registers are overhead: with two sets a wave holds 128 registers for 8 KB
// in flight, and three such waves per SIMD (three admitted K0s) saturate HBM -- 384 registers.  With NS = 4 sets ONE wave holds
// 192 registers for 24 KB in flight: the same bytes in half the registers.  MEASURED AND REJECTED (docs/ROUND_6.md): one K0
// wave per SIMD cannot issue its ~660 instructions per 8 KB fast enough beside four scan waves (0.168 -> 0.24 ms per step with
// one K0 admitted, 0.176-0.19 with two or three) -- the product launches NS = 2; 3 and 4 exist in libporeseg_diag.so.
template <int DT, int NS = 2>
__global__ __launch_bounds__(64 * K0_WAVES, (NS > 2 ? 2 : PS_K0_MINW)) void blocksum_kernel(DevCfg c, const int64_t *__restrict__ ev_start, const int64_t *__restrict__ ev_len,
                                                                const int64_t *__restrict__ ev_boff, int n_ev, int64_t n_samples, void *bs_out,
                                                                int4 *ev_info, int4 *chunk_tot, unsigned *status, uint4 *grp_out)
{
    constexpr bool WIDE = bs_wide<DT>();
    constexpr int LIM = WIDE ? BSW_LIM : BS_WIDE;
    constexpr int ES = static_cast<int>(sizeof(typename Raw<DT>::type));
    constexpr int NV = 8 * ES / 16;                    // 16-byte vectors per block
    __shared__ int4 tr[K0_WAVES][K0_WB];               // per wave: the records of its 256 blocks (transposition)
    const int lane = threadIdx.x & 63;
    const int wave = uni(static_cast<int>(threadIdx.x >> 6));
    const long long nwv = static_cast<long long>(gridDim.x) * K0_WAVES;      // waves of the launch
    long long wv = static_cast<long long>(blockIdx.x) * K0_WAVES + wave;      // this wave's first wave block
    const long long nb_total = ev_boff[n_ev];
    const long long n_wb = nb_total / K0_WB + 1;       // wave blocks with wb0 <= nb_total (the last one holds the end boundary's entry)
    if (wv >= n_wb) return;                            // (waves are independent: no barrier in this kernel)
#if PS_K0_PRIO
    // K0 is the memory-bound kernel of a call: its waves go first when a SIMD arbitrates its vector issue (priority, then age --
    // MI355X_MICROARCH.md), so that what they do between two requests does not queue behind the scan waves of other calls
    __builtin_amdgcn_s_setprio(PS_K0_PRIO);
#endif
    unsigned bad = 0;
    int4 *mine = tr[wave];
    // where a wave block's samples come from (uniform): its event, and whether its 256 blocks are full blocks of that one
    // event (the fast route)
    auto classify = [&](long long wb0, int e_lo) -> K0Src {
        K0Src s;
        const long long gfirst = min(wb0, nb_total - 1);
        int lo = e_lo, hi = n_ev - 1;                  // event e: ev_boff[e] <= gb < ev_boff[e+1]  (wave blocks ascend: e_lo = the previous one's event)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (ev_boff[mid] <= gfirst) lo = mid; else hi = mid - 1;
        }
        s.e = lo;
        s.b_first = wb0 - ev_boff[lo];
        s.base = ev_start[lo];
        const int64_t len_f = ev_len[lo];
        // (no alignment condition since round 5: 16-byte loads from 2- or 4-byte-aligned addresses return the right bytes on
        //  this part at 92 % of the aligned rate -- tools/probes/unaligned_probe.hip -- and the events that a detector cuts
        //  out of an int16 file trace start at any sample: their blocks all took the general route, K0 146 us per 1e8 samples)
        s.fast = wb0 + K0_WB <= ev_boff[lo + 1] && 8 * (s.b_first + K0_WB) <= len_f && nb_total > 0;
        // (a device whose probe failed -- none known -- takes the fast route from 16-byte-aligned addresses only, as until round 4)
        if (!c.k0_unaligned && ((reinterpret_cast<uintptr_t>(c.samples) + static_cast<uintptr_t>((s.base + 8 * s.b_first) * ES)) & 15u) != 0) s.fast = 0;
        return s;
    };
    // the loads of a wave block: issued unconditionally -- a wave block of the general route fetches (and ignores) the first
    // 8 KB of the digest buffer instead -- so that the compiler's count of the loads in flight is the same on every path
    using raw_t = typename Raw<DT>::type;
    auto issue = [&](const K0Src &s, int4 (&raw)[K0_BPT][NV], raw_t &first) {
        // (uniform base + one 32-bit lane offset: the scalar-base form of the loads needs no address registers per row)
        const char *pb = s.fast ? static_cast<const char *>(c.samples) + (s.base + 8 * s.b_first) * ES : static_cast<const char *>(bs_out);
        // (the event's first sample -- the centre m of its sums -- ahead of the block loads: loads return in order, and a load
        //  issued behind them would make the wave wait for the NEXT wave block's samples)
        first = *reinterpret_cast<const raw_t *>(s.fast ? static_cast<const char *>(c.samples) + s.base * ES : static_cast<const char *>(bs_out));
        const unsigned lo = static_cast<unsigned>(lane) * (8u * ES);
#pragma unroll
        for (int k = 0; k < K0_BPT; ++k)
#pragma unroll
            for (int v = 0; v < NV; ++v) raw[k][v] = k0_load16<ES>(pb + (lo + static_cast<unsigned>(k * 64 * 8 * ES + 16 * v)));
    };
    auto process = [&](auto fast_tag, const long long wb0, const K0Src &src, const int4 (&raw)[K0_BPT][NV], const raw_t first) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const int e_first = src.e;
    int ymn_all = 0x7fffffff, ymx_all = -0x7fffffff - 1;   // extremes over the lane's four blocks: the range is checked once
    auto put = [&](int k, const K0Rec<DT> &r) {        // record of block 64 k + lane, in load order
        if (PS_K0_TRIM) { ymn_all = min(ymn_all, r.ymin); ymx_all = max(ymx_all, r.ymax); }
        else if (r.ymax >= LIM || r.ymin <= -LIM) bad |= ST_WIDE_RANGE;
        if constexpr (WIDE) mine[64 * k + lane] = make_int4(r.s1, max(-r.ymin, r.ymax), static_cast<int>(r.s2w), static_cast<int>(r.s2w >> 32));
        else mine[64 * k + lane] = make_int4(r.s1, static_cast<int>(r.s2), (r.ymin & 0xffff) | (r.ymax << 16), 0);
    };
    if constexpr (FAST) {
        const int m = to_count<DT>(c, first, bad);
        const float mf = static_cast<float>(m);        // |m| < 2^23: exact
        unsigned nz = 0;
        // |count| < 2^23 for fp32 input: |k - m| < LIM is checked for every block, so only an m near the limit needs a look
        const bool look = sdt(DT) == PS_DTYPE_F32 && (WIDE || m > 8388607 - LIM || m < -8388607 + LIM);
        // (int16, narrow digest: packed arithmetic on the raw counts; k0 = m - off is the event's first raw count)
        constexpr bool PK16 = sdt(DT) == PS_DTYPE_I16 && !WIDE && PS_K0_PK16;
        const int k0raw = m - c.off_counts;
        const int k0pk = (k0raw & 0xffff) | (k0raw << 16);
#pragma unroll
        for (int k = 0; k < K0_BPT; ++k) {
            int y[8];
            K0Rec<DT> r;
            if constexpr (PK16) {
                k0_block_sums_pk16<DT>(raw[k][0], k0pk, r);
            } else if constexpr (sdt(DT) == PS_DTYPE_F32 && !WIDE && PS_K0_PKF32) {
                k0_block_sums_pkf32<DT>(raw[k], c.inv_q, mf, nz, r);
                asm volatile("" : "+v"(nz));
            } else {
            k0_offsets<DT>(c, raw[k], m, mf, nz, y);
            asm volatile("" : "+v"(nz));               // (the integrality word is complete here: left alone, the compiler keeps x/q and its rounding of all 32 samples for one OR tree at the end)
            k0_block_sums<DT>(y, r);
            }
            if (!PS_K0_TRIM && look) {
                const int ka = m + r.ymin, kb = m + r.ymax;
                if (max(ka < 0 ? -ka : ka, kb < 0 ? -kb : kb) >= 8388608) bad |= ST_OFF_GRID;
            }
            put(k, r);
            __builtin_amdgcn_sched_barrier(0);         // one block after the other: interleaved, the four blocks and the second set of loads do not fit 128 registers
        }
        if (PS_K0_TRIM && look) {
            const int ka = m + ymn_all, kb = m + ymx_all;
            if (max(ka < 0 ? -ka : ka, kb < 0 ? -kb : kb) >= 8388608) bad |= ST_OFF_GRID;
        }
        if (nz) bad |= ST_OFF_GRID;
        if (src.b_first + lane == 0) ev_info[e_first] = make_int4(m, 0, static_cast<int>(ev_boff[e_first] & 0xffffffffLL), static_cast<int>(ev_boff[e_first] >> 32));
    } else {
#pragma unroll 1
        for (int k = 0; k < K0_BPT; ++k) {
            const K0Gen<DT> g = k0_block_general<DT>(bs_cold(c), ev_start, ev_len, ev_boff, n_samples, wb0 + 64 * k + lane, nb_total, e_first, ev_info);
            bad |= g.bad;
            if (sdt(DT) == PS_DTYPE_F32) {
                const int ka = g.m + g.r.ymin, kb = g.m + g.r.ymax;
                if (max(ka < 0 ? -ka : ka, kb < 0 ? -kb : kb) >= 8388608) bad |= ST_OFF_GRID;     // |count| >= 2^23
            }
            put(k, g.r);
        }
    }
    if (PS_K0_TRIM && (ymx_all >= LIM || ymn_all <= -LIM)) bad |= ST_WIDE_RANGE;
    // the records went through LDS in load order (block 64 k + L); read back as blocks 4 L .. 4 L + 3
    k0_lds_ready();                                    // wave-local hand-over (the waves of the workgroup are independent)
    int4 t[K0_BPT];
#pragma unroll
    for (int j = 0; j < K0_BPT; ++j) t[j] = mine[K0_BPT * lane + j];
    const long long gb4 = wb0 + K0_BPT * lane;        // first of this lane's four consecutive blocks
    const long long chunk = (wb0 >> BS_CHUNK_LOG) + (lane >> 5);
    if constexpr (WIDE) {
        // in-thread prefix, then one half-wave scan of the four-block totals (64-bit)
        long long e1[K0_BPT], e2[K0_BPT];
        long long r1 = 0, r2 = 0;
        int yabs = 0;
#pragma unroll
        for (int j = 0; j < K0_BPT; ++j) {
            e1[j] = r1; e2[j] = r2;
            r1 += t[j].x; r2 += i64_of(t[j].z, t[j].w);
            yabs = max(yabs, t[j].y);
        }
        const long long i1 = half_incl_scan_i64(r1), i2 = half_incl_scan_i64(r2);
        const long long x1 = i1 - r1, x2 = i2 - r2;
        int4 *bs = static_cast<int4 *>(bs_out) + gb4;
#pragma unroll
        for (int j = 0; j < K0_BPT; ++j) {
            const long long a1 = x1 + e1[j], a2 = x2 + e2[j];
            bs[j] = make_int4(static_cast<int>(a1), static_cast<int>(a1 >> 32), static_cast<int>(a2), static_cast<int>(a2 >> 32));
        }
        yabs = half_max_i32(yabs);
        if ((lane & 31) == 31) {
            chunk_tot[2 * chunk] = make_int4(static_cast<int>(i1), static_cast<int>(i1 >> 32), static_cast<int>(i2), static_cast<int>(i2 >> 32));
            chunk_tot[2 * chunk + 1] = make_int4(yabs, 0, 0, 0);
        }
    } else {
        int e1[K0_BPT];
        unsigned long long e2[K0_BPT];
        int r1 = 0;
        unsigned long long r2 = 0;                     // four blocks: < 2^33
#pragma unroll
        for (int j = 0; j < K0_BPT; ++j) {
            e1[j] = r1; e2[j] = r2;
            r1 += t[j].x; r2 += static_cast<unsigned>(t[j].y);
        }
        // min / max of the four blocks (packed int16 pairs: low half min, high half max)
        const int mmn = min(min(static_cast<int>(static_cast<short>(t[0].z & 0xffff)), static_cast<int>(static_cast<short>(t[1].z & 0xffff))),
                            min(static_cast<int>(static_cast<short>(t[2].z & 0xffff)), static_cast<int>(static_cast<short>(t[3].z & 0xffff))));
        const int mmx = max(max(t[0].z >> 16, t[1].z >> 16), max(t[2].z >> 16, t[3].z >> 16));
        int yabs = max(-mmn, mmx);
        const int i1 = half_incl_scan_i32(r1);
        const int r2lo = static_cast<int>(r2 & 0xffffu), r2hi = static_cast<int>(r2 >> 16);      // < 2^16, < 2^17: 32 lanes < 2^22
        const int ilo = half_incl_scan_i32(r2lo), ihi = half_incl_scan_i32(r2hi);
        const int x1 = i1 - r1;
        const unsigned long long x2 = (static_cast<unsigned long long>(static_cast<unsigned>(ihi - r2hi)) << 16) + static_cast<unsigned>(ilo - r2lo);
        uint2 ent[K0_BPT];
#pragma unroll
        for (int j = 0; j < K0_BPT; ++j) ent[j] = bs8_pack(x1 + e1[j], x2 + e2[j]);
        uint4 *bs = reinterpret_cast<uint4 *>(static_cast<uint2 *>(bs_out) + gb4);
        bs[0] = make_uint4(ent[0].x, ent[0].y, ent[1].x, ent[1].y);
        bs[1] = make_uint4(ent[2].x, ent[2].y, ent[3].x, ent[3].y);
        if (c.blk_mm) *reinterpret_cast<int4 *>(c.blk_mm + gb4) = make_int4(t[0].z, t[1].z, t[2].z, t[3].z);
        if (c.blk_cls) {
            // single-pass file route (the call's one event is the whole trace): the detector's verdict on this lane's four blocks,
            // 2 bits each, and the extremes of the blocks inside the trace per half wave (seg_device.hpp: edge_cls_kernel)
            unsigned b2 = 0;
            const int ythr = c.cls_kthr - load_count<DT>(c, ev_start[0], b2);
            unsigned cb = 0;
            int cmn = 0x7fffffff, cmx = -0x7fffffff;
#pragma unroll
            for (int j = 0; j < K0_BPT; ++j) {
                const int lo_j = static_cast<int>(static_cast<short>(t[j].z & 0xffff)), hi_j = t[j].z >> 16;
                cb |= (hi_j < ythr ? 1u : lo_j >= ythr ? 0u : 2u) << (2 * j);
                if (gb4 + j < nb_total) { cmn = min(cmn, lo_j); cmx = max(cmx, hi_j); }
            }
            c.blk_cls[gb4 >> 2] = static_cast<unsigned char>(cb);
            cmn = -half_max_i32(-cmn);
            cmx = half_max_i32(cmx);
            if ((lane & 31) == 31) c.cls_mm[chunk] = make_int2(cmn, cmx);
        }
#if PS_K0_AMP == 2
        if (grp_out) {
            // group record (round 4, second form): the eight lanes 8 g .. 8 g + 7 hold the group's 32 blocks.  Per block b the sum of
            // squares ABOUT THE GROUP'S MEAN, q_b = s2_b - mu_g (2 s1_b - 8 mu_g), is formed exactly (fp64: s2 < 2^31, the
            // product a multiple of 2^-13 below 2^33) and rounded to fp32 once; everything after it is fp32 on small numbers:
            //     d1 at the block boundaries: 32 d1 = 32 P1 - nb S1g in int32 (P1: sum of the group's first nb blocks),
            //     d2z at the block boundaries: prefix of (q_b - SS_g / 32), lane-local plus an exclusive scan over the octet,
            // and inside block b the bridges leave the chord between its two boundaries by at most
            //     |w1| <= sqrt(2 q_b)     (Cauchy-Schwarz on the block-centred samples; q_b is at least their sum of squares)
            //     |w2| <= 7/8 q_b         (the squares are non-negative: the partial sum of j of 8 lies in [-(j/8) q_b, (1 - j/8) q_b])
            // so  D1 = max_b (max(|d1(8b)|, |d1(8b+8)|) + sqrt(2 q_b)),   D2 = max_b (max(|d2z(8b)|, |d2z(8b+8)|) + 7/8 q_b).
            // Rounding: every q_b carries 2^-24 relative, the prefixes at most 2^-18 SS_g in all: D2 takes 1e-5 SS_g on top.
            const int S1g = oct_allsum(r1);
            const double mg = static_cast<double>(S1g) * (-1.0 / 8192.0);                       // -(mu_g / 32)
            float qf[K0_BPT];
#pragma unroll
            for (int j = 0; j < K0_BPT; ++j) {
                const int U = 64 * t[j].x - S1g;                                                // 32 (2 s1_b - 8 mu_g), |U| < 2^24
                const float qx = static_cast<float>(fma(mg, static_cast<double>(U), static_cast<double>(static_cast<unsigned>(t[j].y))));
                qf[j] = PS_K0_TRIM ? __int_as_float(max(__float_as_int(qx), 0)) : fmaxf(qx, 0.0f);     // (a negative float is a negative integer)
            }
            const float lq = (qf[0] + qf[1]) + (qf[2] + qf[3]);
            float SSg = lq;
#define PS_STEP(CTRL) { SSg += dpp_movf<CTRL, 0xf>(0.0f, SSg); }
            PS_OCT_STEPS(PS_STEP)
#undef PS_STEP
            const float qbar = SSg * (1.0f / static_cast<float>(BS_GRP));
            // exclusive scan of the lanes' (sum of q_b - qbar) over the octet: row_shr 1, 2, 4, lanes nearer than that to the octet's start masked
            const float lw = fmaf(-static_cast<float>(K0_BPT), qbar, lq);
            float inc = lw;
            const int l7 = lane & 7;
            { const float v = dpp_movf<0x111, 0xf>(0.0f, inc); inc += l7 >= 1 ? v : 0.0f; }
            { const float v = dpp_movf<0x112, 0xf>(0.0f, inc); inc += l7 >= 2 ? v : 0.0f; }
            { const float v = dpp_movf<0x114, 0xf>(0.0f, inc); inc += l7 >= 4 ? v : 0.0f; }
            float d2 = inc - lw;                                                                // d2z at this lane's first block boundary
            int p1 = x1 - oct_first(x1, lane);                                                  // P1 at this lane's first block
            int nb = K0_BPT * l7;
            float a1p = fabsf(static_cast<float>(32 * p1 - nb * S1g)), a2p = fabsf(d2);
            float D1 = 0.0f, D2 = 0.0f;
#pragma unroll
            for (int j = 0; j < K0_BPT; ++j) {
                p1 += t[j].x; nb += 1;
                d2 += qf[j] - qbar;
                const float a1n = fabsf(static_cast<float>(32 * p1 - nb * S1g)), a2n = fabsf(d2);
#if PS_K0_TRIM
                // (all operands are non-negative and finite: their maxima are integer maxima of the bit patterns -- no quieting
                //  moves in front of v_max_f32; sqrt(2 q) 1.00001 = sqrt(q) * 1.4142278)
                D1 = pmaxf(D1, fmaf(pmaxf(a1p, a1n), 1.0f / 32.0f, __builtin_amdgcn_sqrtf(qf[j]) * 1.4142278f));
                D2 = pmaxf(D2, fmaf(0.875f, qf[j], pmaxf(a2p, a2n)));
#else
                D1 = fmaxf(D1, fmaf(fmaxf(a1p, a1n), 1.0f / 32.0f, __builtin_amdgcn_sqrtf(2.0f * qf[j]) * 1.00001f));
                D2 = fmaxf(D2, fmaf(0.875f, qf[j], fmaxf(a2p, a2n)));
#endif
                a1p = a1n; a2p = a2n;
            }
            D1 = oct_allmaxf(D1) * 1.00001f;
            D2 = fmaf(1.0e-5f, SSg, oct_allmaxf(D2) * 1.00001f);
            if (l7 == 0) grp_out[gb4 >> BS_GRP_LOG] = make_uint4(ent[0].x, ent[0].y, __float_as_uint(D1), __float_as_uint(D2));
        }
#else
        if (grp_out) {
            // group record: the eight lanes 8 g .. 8 g + 7 hold the group's 32 blocks.  Totals by an all-reduce, the prefix
            // at the group's first block from its first lane; then every lane looks at the boundaries behind its four blocks:
            //     32 d1 = 32 P1 - jb S1g                          (int32: |P1| < 2^22)
            //     d2z   = P2 - jb S2g / 32 - S1g d1 / 128         (fp64: exact integers / 4096 below 2^50)
            // (P1, P2: sums of the group's first jb blocks).  The last boundary of the group gives 0 by itself.
            const int S1g = oct_allsum(r1);
            const int S2lo = oct_allsum(r2lo), S2hi = oct_allsum(r2hi);                          // < 2^19, < 2^20
            const double S2g = static_cast<double>(S2hi) * 65536.0 + static_cast<double>(S2lo);
            const int xlo = ilo - r2lo, xhi = ihi - r2hi;
            int p1 = x1 - oct_first(x1, lane);
            double p2 = static_cast<double>(xhi - oct_first(xhi, lane)) * 65536.0 + static_cast<double>(xlo - oct_first(xlo, lane));
            const double S1gd = static_cast<double>(S1g);
            const double k2 = S2g * (-1.0 / 32.0), k1 = S1gd * (-1.0 / 4096.0);
            const int jb0 = K0_BPT * (lane & 7);
            int m1 = 0;
            double m2 = 0.0;
#pragma unroll
            for (int j = 0; j < K0_BPT; ++j) {
                p1 += t[j].x;
                p2 += static_cast<double>(static_cast<unsigned>(t[j].y));
                const int jb = jb0 + j + 1;
                const int d1x = 32 * p1 - jb * S1g;
                const double d2 = fma(k1, static_cast<double>(d1x), fma(static_cast<double>(jb), k2, p2));
                m1 = max(m1, d1x < 0 ? -d1x : d1x);
                m2 = fmax(m2, fabs(d2));
            }
            const int gmn = oct_allmin(mmn), gmx = oct_allmax(mmx);
            m1 = oct_allmax(m1);
            const float m2f = oct_allmaxf(static_cast<float>(m2) * 1.000001f);
            const float mug = static_cast<float>(S1g) * (1.0f / static_cast<float>(BS_GRP_SAMPLES));
            const float Mg = fmaxf(static_cast<float>(gmx) - mug, mug - static_cast<float>(gmn)) * 1.000001f + 0.01f;
            const float D1 = (static_cast<float>(m1) * (1.0f / 32.0f) + 2.0f * static_cast<float>(gmx - gmn)) * 1.000001f;
            const float D2 = fmaf(2.0f * Mg, Mg, m2f) * 1.000001f;
            if ((lane & 7) == 0) grp_out[gb4 >> BS_GRP_LOG] = make_uint4(ent[0].x, ent[0].y, __float_as_uint(D1), __float_as_uint(D2));
        }
#endif
        yabs = half_max_i32(yabs);
        if ((lane & 31) == 31) {
            const unsigned long long tot2 = (static_cast<unsigned long long>(static_cast<unsigned>(ihi)) << 16) + static_cast<unsigned>(ilo);
            chunk_tot[chunk] = make_int4(i1, yabs, static_cast<int>(tot2), static_cast<int>(tot2 >> 32));
        }
    }
    k0_lds_done();                                     // the next wave block's records may overwrite the transposition buffer
    };
    // Pass 1, the wave blocks of the fast route: two sets of load registers, used in turn -- the next wave block's samples
    // travel while this one is worked on.  Wave blocks of the general route (an out-of-line call: registers that are live
    // across it would be spilled, and the compiler then parks the whole second set on the stack) are left to pass 2.
    int4 raw[NS][K0_BPT][NV];
    raw_t first[NS];
    K0Src src[NS];
    unsigned long long general = 0;                    // iterations of this wave that were not fast (uniform; from the 64th on: looked at again)
    int it = 0;
    const long long wv0 = wv;
    // (a wave block that is skipped here still "reads" its registers: on every path the loads of a set are then known to be
    //  complete before the set is requested again -- otherwise the compiler waits in the middle of the next request)
    auto retire = [&](const int4 (&rw)[K0_BPT][NV], const raw_t fs) {
#pragma unroll
        for (int k = 0; k < K0_BPT; ++k)
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("" :: "v"(rw[k][v].x), "v"(rw[k][v].y), "v"(rw[k][v].z), "v"(rw[k][v].w));
        asm volatile("" :: "v"(fs));
    };
    // prologue: sets 0 .. NS-2 hold this wave's first NS-1 wave blocks (wv, wv + nwv, ...); set NS-1 is requested in the loop
    int e_hint = 0;
#pragma unroll
    for (int q = 0; q < NS - 1; ++q) {
        const long long wq = wv + q * nwv;
        if (wq < n_wb) { src[q] = classify(wq * K0_WB, e_hint); e_hint = src[q].e; }
        else { src[q] = src[0]; src[q].fast = 0; }     // (beyond the call: the dummy lines)
        issue(src[q], raw[q], first[q]);
    }
    bool last = false;
#pragma unroll 1
    while (!last) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            if (last) break;
            constexpr int NSm1 = NS - 1;
            const int f = (q + NSm1) % NS;              // the set that was worked on last: free again (compile-time after unrolling)
            const long long wn = wv + static_cast<long long>(NSm1) * nwv;
            if (wn < n_wb) { src[f] = classify(wn * K0_WB, e_hint); e_hint = src[f].e; } else src[f].fast = 0;
            issue(src[f], raw[f], first[f]);            // (unconditional, like the loads inside: beyond the call the dummy lines)
            if (src[q].fast) process(std::true_type{}, wv * K0_WB, src[q], raw[q], first[q]);
            else { general |= 1ull << min(it, 63); retire(raw[q], first[q]); }
            if (wv + nwv >= n_wb) last = true;
            else { wv += nwv; ++it; }
        }
    }
    // Pass 2, the general route: the wave blocks at the ends of events, unaligned events, the last wave block of the call
    if (general) {
        int e_lo = 0;
        it = 0;
#pragma unroll 1
        for (wv = wv0; wv < n_wb; wv += nwv, ++it) {
            if (it < 63 && !((general >> it) & 1ull)) continue;
            const K0Src sg = classify(wv * K0_WB, e_lo);
            e_lo = sg.e;
            if (!sg.fast) process(std::false_type{}, wv * K0_WB, sg, raw[0], first[0]);
        }
    }
    if (bad) atomicOr(status, bad);
}

// ---- screen arithmetic from window-relative sums ------------------------------------------------------
struct BsEval { float g; f2 lg; f2 r; f2 u; bool okL, okR; };

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
    o.u = u;
    o.okL = u.x >= vfloor;
    o.okR = u.y >= vfloor;
    const f2 t = nv * o.lg;
    o.g = -(t.x + t.y);
    return o;
}


// Upper bound of the screened gain of the 7 candidates inside the block (p, q), q = p + 8, from the evaluations of BOTH of
// its boundaries (ep at p, eq at q) -- the corner bound of the sweep takes SS_L at p and SS_R at q together, which no
// candidate has (tools/experiments/two_boundary_bound.py checks this form against the exact gains).
// With SS the sums of squared deviations of the two parts and B(k) = N(k)^2 / (n k (n - k)), N(k) = n S1_L(k) - k T1:
//     SS_L(k) + SS_R(k) = SS_tot - B(k)      exactly;
// inside the block SS_L grows from SS_L(p) to SS_L(q), SS_R falls from SS_R(p) to SS_R(q), and
//     |N(k)| <= min(|N(p)|, |N(q)|) + n sqrt(7 Q),   Q = sum over the block of (y - T1/n)^2      (Cauchy-Schwarz)
//                                                      <= 2 (q/p) (SS_L(q) - SS_L(p)) + 16 B(p) (n - p) / (n p),
// so (SS_L, SS_R) lies in that box on or above the line SS_L + SS_R = SS_tot - Bmax.  The cost
// k log(SS_L/k) + (n-k) log(SS_R/(n-k)) increases in both arguments and is concave along the line and in k: its minimum is
// next to one of the boundaries --
//     gain(k) <= max( P(p) + (n-p) e1 ,  P(q) + q e2 ),    P = gain + 7 |lgL - lgR| + 49 log2e (1/nl + 1/nr),
//     e1 = log2e s1 / (SS_R(p) - s1),  s1 = Bmax - B(p);    e2 = log2e s2 / (SS_L(q) - s2),  s2 = Bmax - B(q).
// On noise the corner bound lies 10.7 (median) / 28.5 (99 %) log2 units above the larger boundary gain, this one 0.8 / 10.9.
// Returns +inf when one of the four variances is below the floor.
__device__ __forceinline__ float bs_block_bound2(const BsEval &ep, const BsEval &eq, int p, int n, float SSt, float rn)
{
    const float LOG2E = 1.4426950408889634f;
    const float nf = static_cast<float>(n);
    const float pf = static_cast<float>(p), qf = pf + 8.0f, npf = nf - pf, nqf = npf - 8.0f;
    const float SSLp = pf * ep.u.x, SSRp = npf * ep.u.y, SSLq = qf * eq.u.x, SSRq = nqf * eq.u.y;
    const float eB = SSt * 2.0e-6f;                                          // rounding of the fp32 sums of squares
    const float Bp = (SSt - SSLp) - SSRp, Bq = (SSt - SSLq) - SSRq;
    const float Bpu = fmaxf(Bp, 0.0f) + eB, Bqu = fmaxf(Bq, 0.0f) + eB;
    const float DLb = fmaxf(SSLq - SSLp, 0.0f) + eB;
    const float Qb = fmaf(2.0f * fmaf(8.0f, ep.r.x, 1.0f), DLb, 16.0f * Bpu * (npf * rn) * ep.r.x);
    const float rK = __builtin_amdgcn_rcpf(fminf(pf * npf, qf * nqf));
    const float kap = fmaf(8.0f, fmaxf(ep.r.x, eq.r.y), 1.0f);
    const float sBQ = __builtin_amdgcn_sqrtf(fminf(Bpu, Bqu) * kap) + __builtin_amdgcn_sqrtf(7.0f * nf * Qb * rK);
    const float Bm = fmaf(sBQ, sBQ * 1.00001f, eB);
    const float s1 = fmaxf(Bm - (Bp - eB), 0.0f), s2 = fmaxf(Bm - (Bq - eB), 0.0f);
    const float v1 = fmaxf(SSRp - s1, SSRq), u2 = fmaxf(SSLq - s2, SSLp);
    const float e1 = (LOG2E * 1.00001f) * (SSRp - v1) * __builtin_amdgcn_rcpf(v1);
    const float e2 = (LOG2E * 1.00001f) * (SSLq - u2) * __builtin_amdgcn_rcpf(u2);
    const float Pp = ep.g + fmaf(7.0f, fabsf(ep.lg.x - ep.lg.y), (49.0f * LOG2E) * (ep.r.x + ep.r.y));
    const float Pq = eq.g + fmaf(7.0f, fabsf(eq.lg.x - eq.lg.y), (49.0f * LOG2E) * (eq.r.x + eq.r.y));
    const float hb = fmaxf(fmaf(npf, e1, Pp), fmaf(qf, e2, Pq)) + 1.0e-3f;
    const bool sound = static_cast<bool>(static_cast<int>(ep.okL) & static_cast<int>(ep.okR) & static_cast<int>(eq.okL) & static_cast<int>(eq.okR));
    return sound ? hb : INFINITY;
}

// What the gain can rise by, inside a group next to the boundary evaluated as `e`, over the gain on the chord between the
// group's two boundaries (log2 units; +inf when the expansion does not apply).  Derivation (DESIGN.md 4.4,
// tools/experiments/group_bound.py): the gain is a convex function of the left part's (k, S1, S2) -- k phi(S1/k, S2/k) with
// phi(a, b) = log(b - a^2) is the perspective of a concave function --, the prefix path of the group lies in the
// parallelepiped  chord(k) + (0, d1, d2z + 2 mu_g d1),  |d1| <= D1, |d2z| <= D2  (the amplitudes K0 left in the group
// record), so the gain on the path is at most the largest gain at the vertices: k at one of the two boundaries, (d1, d2z) at
// the corners of the box.  At a boundary the displacement changes the sums of squared deviations by
//     dSS_L = d2z + 2 wL d1 - d1^2 / k,        dSS_R = -d2z - 2 wR d1 - d1^2 / (n-k),      wL = mu_g - mu_L,  wR = mu_g - mu_R,
// and with log(1 + t) >= t - c(T) t^2 for |t| <= T the cost k log(SS_L/k) + (n-k) log(SS_R/(n-k)) falls by at most
//     D2 |1/V_L - 1/V_R| + 2 D1 |wL/V_L - wR/V_R| + D1^2 (1/SS_L + 1/SS_R) + c(tL) k tL^2 + c(tR) (n-k) tR^2,
//     tL = (D2 + 2 |wL| D1 + D1^2 / k) / SS_L,  tR alike.
// c(T) = sup over |t| <= T of (t - log(1 + t)) / t^2 = (-T - log(1 - T)) / T^2 = 1/2 + T/3 + T^2/4 + ... is convex in T, so
// on 0 <= T <= 0.3 it lies below its chord 0.5 + 0.4325 T (c(0.3) = 0.62973); the kernels take 0.5 + 0.44 T and give up
// beyond T = 0.3.  (Until the middle of round 4: the constant 0.54 and T <= 0.09 -- which refused the first full group
// at either end of nearly every window and the second one in half of them, k being a few hundred there: 3.6 live rows in a
// window without a split, 2.4 with the wider range.)
// The first-order terms keep the cancellation between the two sides (on noise 1/V_L - 1/V_R is ~ sqrt(2/k) / sigma^2).
// wa, wb: mu_g - mu_L for the two neighbouring groups (the same value twice at the ends); dmu = mu_L - mu_R, so that
// wR = wL + dmu and wL/V_L - wR/V_R = wL (1/V_L - 1/V_R) - dmu / V_R.  All inputs carry relative errors of a few 2^-24
// (exact fp64 numerators), the result is inflated by 1e-4 + 1e-3.
__device__ __forceinline__ float bs_group_slack(const BsEval &e, float D1, float D2, float wa, float wb, float dmu)
{
    const float LOG2E = 1.4426950408889634f;
    const float iVL = __builtin_amdgcn_rcpf(e.u.x), iVR = __builtin_amdgcn_rcpf(e.u.y);
    const float gS = iVL - iVR, dV = dmu * iVR;
    const float c1 = fmaxf(fabsf(fmaf(wa, gS, -dV)), fabsf(fmaf(wb, gS, -dV)));
    const float wL = fmaxf(fabsf(wa), fabsf(wb)), wR = fmaxf(fabsf(wa + dmu), fabsf(wb + dmu));
    const float D11 = D1 * D1;
    const float iSL = iVL * e.r.x, iSR = iVR * e.r.y;                        // 1 / SS_L, 1 / SS_R
    const float first = fmaf(D2, fabsf(gS), fmaf(2.0f * D1, c1, D11 * (iSL + iSR)));
    const float EL = fmaf(2.0f * wL, D1, fmaf(D11, e.r.x, D2)), ER = fmaf(2.0f * wR, D1, fmaf(D11, e.r.y, D2));
    const float tL = EL * iSL, tR = ER * iSR;
    const float second = fmaf(fmaf(0.44f, tL, 0.5f) * (EL * iVL), tL, fmaf(0.44f, tR, 0.5f) * (ER * iVR) * tR);   // k tL^2 = EL^2 / (k V_L^2)
    const float A = fmaf((LOG2E * 1.0001f), first + second, 1.0e-3f);
    return (tL <= 0.3f && tR <= 0.3f) ? A : INFINITY;                       // (NaN compares false: +inf)
}

struct BsQ { int j, a1; double a2; };                     // queued block (J-8, J): its end J, sums of [ps, J)
struct BsC { int j, a1; double a2; float g; int pad; };   // contender: candidate, its exact sums, screened gain
// the same for the wide digest (64-bit integer sums; |S1| < 2^40 shares a word with the window-relative position)
struct BsQW { long long ja, a2; };                        // ja = S1 * 2^18 + (J - ps)
struct BsCW { int j; float g; long long a1, a2; };
static_assert(sizeof(BsQW) == sizeof(BsQ) && sizeof(BsCW) == sizeof(BsC), "LDS layout shared by both digests");
template <bool WIDE> struct BsTypes { typedef int s1_t; typedef double s2_t; typedef BsQ Q; typedef BsC C; typedef uint2 E; typedef double o2_t; };
template <> struct BsTypes<true> { typedef long long s1_t; typedef long long s2_t; typedef BsQW Q; typedef BsCW C; typedef int4 E; typedef long long o2_t; };
__device__ __forceinline__ double bs_d(int x) { return static_cast<double>(x); }
__device__ __forceinline__ double bs_d(double x) { return x; }
__device__ __forceinline__ double bs_d(long long x) { return d_of_i64(x); }
__device__ __forceinline__ void bs_q_put(BsQ &q, int J, int ps, int a1, double a2) { q.j = J; q.a1 = a1; q.a2 = a2; }
__device__ __forceinline__ void bs_q_put(BsQW &q, int J, int ps, long long a1, long long a2) { q.ja = a1 * 262144LL + (J - ps); q.a2 = a2; }
__device__ __forceinline__ int bs_q_j(const BsQ &q, int ps) { return q.j; }
__device__ __forceinline__ int bs_q_j(const BsQW &q, int ps) { return ps + static_cast<int>(q.ja & 262143LL); }
__device__ __forceinline__ int bs_q_a1(const BsQ &q) { return q.a1; }
__device__ __forceinline__ long long bs_q_a1(const BsQW &q) { return q.ja >> 18; }
// exact sums of [ps, J) at a digest entry: entry + the offset of its chunk.  Narrow digest: the second moment comes
// out of the entry as the double 2^52 + E2 (bit pattern), the chunk offset carries -2^52: one exact fp64 addition.
__device__ __forceinline__ int bs_a1(uint2 e, int o1) { return bs8_s1(e) + o1; }
__device__ __forceinline__ double bs_a2(uint2 e, double k2) { return bs8_s2_biased(e) + k2; }
__device__ __forceinline__ long long bs_a1(const int4 &e, long long o1) { return i64_of(e.x, e.y) + o1; }
__device__ __forceinline__ long long bs_a2(const int4 &e, long long o2) { return i64_of(e.z, e.w) + o2; }
// count of one sample WITHOUT the grid check: every sample of an event was validated by K0 before a scan reads it
template <int DT>
__device__ __forceinline__ int bs_count_of(const DevCfg &c, typename Raw<DT>::type raw)
{
    if (sdt(DT) == PS_DTYPE_F32) return __float2int_rn(static_cast<float>(raw) * c.inv_q);
    return static_cast<int>(raw) + c.off_counts;
}
template <int DT>
__device__ __forceinline__ int bs_count(const DevCfg &c, int64_t gi)
{
    return bs_count_of<DT>(c, static_cast<const typename Raw<DT>::type *>(c.samples)[gi]);
}
constexpr int BS_NC = 64;                                 // contenders kept per window
#ifndef PS_ONE_ROW_LOOP
#define PS_ONE_ROW_LOOP 0                                 // 1: windows that sweep every row use the loop of the windows with a row mask (one copy of the row code)
#endif
#ifndef PS_RING_GUARD
#define PS_RING_GUARD 0                                   // 1: a slot of the ring is refilled only when a live row is left (no request for row 0 that nobody reads)
#endif
#ifndef PS_BS_D
#define PS_BS_D 2
#endif
constexpr int BS_D = PS_BS_D;                             // rows in flight: a ring of digest entries in registers (8 bytes each), row r + BS_D
                                                          // is requested when row r has been evaluated -- the prefetch distance of a lone chain
constexpr int BS_QN = 128;                                // queued blocks: a drain starts beyond BS_EARLY, and a row adds at most 63
#ifndef PS_DRAIN_FILTER
#define PS_DRAIN_FILTER 1                                 // queued blocks are re-judged from both of their boundaries before their samples are fetched (bs_block_bound2)
#endif
#ifndef PS_TREE_SAMPLE
#ifndef PS_ROW_OFFSETS_BPERMUTE
#define PS_ROW_OFFSETS_BPERMUTE 1
#endif
__device__ __forceinline__ int bs_from_lane(int x, int byte_index) { return __builtin_amdgcn_ds_bpermute(byte_index, x); }
__device__ __forceinline__ long long bs_from_lane(long long x, int byte_index)
{
    const int lo = __builtin_amdgcn_ds_bpermute(byte_index, static_cast<int>(x)), hi = __builtin_amdgcn_ds_bpermute(byte_index, static_cast<int>(x >> 32));
    return (static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo);
}
__device__ __forceinline__ double bs_from_lane(double x, int byte_index) { return __longlong_as_double(bs_from_lane(__double_as_longlong(x), byte_index)); }
#define PS_TREE_SAMPLE 0                                  // 1: subtree windows run the sampling pass too (measured: see scan_window_bs)
#endif
#ifndef PS_EDGE_PRELOAD
#define PS_EDGE_PRELOAD 1                                 // windows with a coarse pass request the rows of their two ends with the setup loads
#endif
constexpr int BS_EARLY = 64;                              // ... and started early beyond this many (<= BS_QN - 64)
constexpr int BS_STRIDE = 63;                             // new boundaries per row (lane 0 repeats the previous row's last)
constexpr int BS_LDS_BYTES = (static_cast<int>(sizeof(BsQ)) + 8) * BS_QN + static_cast<int>(sizeof(BsC)) * BS_NC + 64 * 32;   // (+ 8: the block's own sums)
static_assert(sizeof(QEnt) * SharedT<64>::QN >= BS_LDS_BYTES, "SharedT<64>::q too small");   // (64 * 32: staged blocks of the wide digest, 8 int32 each)

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
// sum over lanes 8 j .. 8 j + 7 of each group of eight, in lane 8 j + 7 (three row_shr steps)
__device__ __forceinline__ int oct_sum(int x)
{
    x += dpp_mov<0x111, 0xf>(0, x); x += dpp_mov<0x112, 0xf>(0, x); x += dpp_mov<0x114, 0xf>(0, x);
    return x;
}
__device__ __forceinline__ long long oct_sum(long long x)
{
#define PS_STEP(CTRL) { const int lo_ = dpp_mov<CTRL, 0xf>(0, static_cast<int>(x)), hi_ = dpp_mov<CTRL, 0xf>(0, static_cast<int>(x >> 32)); x += i64_of(lo_, hi_); }
    PS_STEP(0x111) PS_STEP(0x112) PS_STEP(0x114)
#undef PS_STEP
    return x;
}

// ---- cold paths of the scan, out of line (they hold the fp64 logarithms and divisions: inlined they would set the
//      register count of the whole kernel) --------------------------------------------------------------------------
template <int DT>
__device__ __attribute__((noinline)) long long bs_scan_exact(BsCold k, int64_t g0, int ps, int n, int cand_lo, int cand_hi,
                                                             double thresh, SharedT<64> *sh)
{
    const DevCfg c = bs_cold_cfg(k);
    unsigned bad = 0;
    const int r = scan_exact<64, DT>(c, nullptr, g0, ps, n, cand_lo, cand_hi, thresh, nullptr, *sh, bad, nullptr);
    return (static_cast<long long>(bad) << 32) | static_cast<unsigned>(r);
}

// The contenders (at most 64, one per lane) decided with the reference's fp64 arithmetic; first maximum wins.
template <int DT>
__device__ __attribute__((noinline)) long long bs_decide(BsCold k, int m, const void *cont_v, int ccount, int ps, int n,
                                                         typename BsTypes<bs_wide<DT>()>::s1_t T1, typename BsTypes<bs_wide<DT>()>::s2_t T2,
                                                         double thresh)
{
    const DevCfg cfg = bs_cold_cfg(k);
    const DevCfg *c = &cfg;
    int near = 0;
    constexpr bool WIDE = bs_wide<DT>();
    typedef typename BsTypes<WIDE>::C BsC_t;
    const BsC_t *cont = static_cast<const BsC_t *>(cont_v);
    const int lane = threadIdx.x & 63;
    const double T1d = bs_d(T1), T2d = bs_d(T2), dn = static_cast<double>(n);
    double var_summed;
    if constexpr (WIDE) var_summed = dn * log(ref_var(T1d, T2d, n, c->q, c->q2));
    else var_summed = dn * log(ref_var(T1d + dn * static_cast<double>(m),
                                       T2d + 2.0 * static_cast<double>(m) * T1d + dn * static_cast<double>(m) * static_cast<double>(m), n, c->q, c->q2));
    double eg = thresh;
    int ei = -1;
    double gx = -INFINITY;
    // Wide digest = a re-quantised float64 current (a filtered event, data on no grid).  The reference segments the
    // UNROUNDED values with sequential fp64 cumsums (cparsers.pyx:110-111), whose rounding noise in a part of n_p samples
    // that ends at sample i of the event is about ulp(c2[i]) sqrt(n_p / 12) in its sum of squares, i.e.
    // ulp(c2[i]) sqrt(n_p / 12) / V_p in its term n_p log V_p of a gain; c2[i] ~ i * level^2 (level: the DC level the caller
    // names as offset_counts, plus m).  The rounding of the samples to the grid adds ~ 0.6 sqrt(n_p) / sigma_p.  `noise` is
    // that estimate for this lane's candidate (left + right part), noise_tot the whole window's: decisions whose margin lies
    // within four times the sum of the two sides' estimates are counted as near ties (below) -- on this route the device
    // and the reference can place such a boundary one sample apart (241 per 1e6 boundaries of heavily
    // over-segmented filtered events, none where the segmenter is given the filter's cutoff).
    double noise = 0.0, noise_tot = 0.0;
    const double nk = static_cast<double>(k.noise_k);
    if (lane < ccount) {
        const BsC_t e = cont[lane];
        if constexpr (WIDE) {
            const int nl = e.j - ps;
            const double vl = ref_var(d_of_i64(e.a1), d_of_i64(e.a2), nl, c->q, c->q2);
            const double vr = ref_var(d_of_i64(T1 - e.a1), d_of_i64(T2 - e.a2), n - nl, c->q, c->q2);
            gx = ref_gain(var_summed, nl, vl, n - nl, vr);
            const double vt = ref_var(T1d, T2d, n, c->q, c->q2);
            const double lvl = (fabs(k.dc_counts + static_cast<double>(m)) + sqrt(fmax(vt, 0.0) / c->q2)) * c->q;     // pA
            const double ulp2 = static_cast<double>(ps + n) * lvl * lvl * 2.220446049250313e-16;                    // ulp of c2 at the window's end
            const double tiny = 1.0e-300;
            auto part = [&](int np, double vp) {
                const double sn = sqrt(static_cast<double>(np));
                return ulp2 * sn * 0.2887 / fmax(vp, tiny) + 0.6 * sn * c->q / sqrt(fmax(vp, tiny));
            };
            noise = part(nl, vl) + part(n - nl, vr);
            noise_tot = part(n, vt);
        } else {
            gx = bs_exact_gain(*c, m, e.a1, e.a2, T1, T2, e.j - ps, n, var_summed);
        }
        if (gx > eg) { eg = gx; ei = e.j; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const double og = __shfl_xor(eg, d);
        const int oi = __shfl_xor(ei, d);
        if (beats(og, oi, eg, ei)) { eg = og; ei = oi; }
    }
    // Near-tie accounting: the device logarithm is not glibc's bit for bit, gains carry ~1e-11 of absolute
    // error.  A decision whose margin -- winner against the best other contender, or against the threshold -- lies inside
    // 1e-9 * max(1, |gain|) could have gone the other way in the reference: counted, reported by ps_get_timings.
    if (ei >= 0) {
        double other = (lane < ccount && cont[lane].j != ei) ? gx : -INFINITY;      // NaN gains never win and never tie
        if (!(other == other)) other = -INFINITY;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) other = fmax(other, __shfl_xor(other, d));
        const double tol = 1.0e-9 * fmax(1.0, fabs(eg));
        near = (eg - other < tol || eg - thresh < tol) ? 1 : 0;
        if constexpr (WIDE) {
            // winner's own estimate to every lane, then lane by lane: is this contender within the noise of the winner?
            double nw = (lane < ccount && cont[lane].j == ei) ? noise : 0.0;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) nw = fmax(nw, __shfl_xor(nw, d));
            const bool mine = lane < ccount && cont[lane].j != ei && gx == gx && eg - gx < nk * (nw + noise);
            const bool thr_near = eg - thresh < nk * (nw + __shfl(noise_tot, 0));
            if (__ballot(mine) != 0ull || thr_near) near = 1;
        }
    } else {
        double best = (lane < ccount && gx == gx) ? gx : -INFINITY;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) best = fmax(best, __shfl_xor(best, d));
        const double tol = 1.0e-9 * fmax(1.0, fabs(thresh));
        near = thresh - best < tol ? 1 : 0;
        if constexpr (WIDE) {
            const bool mine = lane < ccount && gx == gx && thresh - gx < nk * (noise + noise_tot);
            if (__ballot(mine) != 0ull) near = 1;
        }
    }
    return (static_cast<long long>(near) << 32) | static_cast<unsigned>(ei);
}

// audit record (c.dbg, 12 words): [0] blocks with a corner bound, [1] violations, [2] blocks with a two-boundary bound,
// [3] violations, [4] groups with a bound, [5] violations, [6..8] smallest margin (bound - largest covered gain) per kind
// as order-preserving int32, [9] windows.  A violation: a covered candidate's screened gain exceeds the bound by more
// than `tol` (2 delta: the slack every pruning level carries).
__device__ __forceinline__ void bs_audit_note(unsigned long long *a, int icnt, int iviol, int imin, float bound, float gmax, float tol)
{
    if (!(bound < INFINITY)) return;                   // no bound was claimed
    atomicAdd(&a[icnt], 1ull);
    if (gmax == -INFINITY) return;
    const float margin = bound - gmax;
    if (!(margin >= -tol)) atomicAdd(&a[iviol], 1ull);
    int k = __float_as_int(margin);
    k ^= (k >> 31) & 0x7fffffff;
    atomicMin(reinterpret_cast<int *>(&a[imin]), k);
}

#ifdef PS_MARKS
#define PS_MARK(n) asm volatile("; PSMARK " #n)
#else
#define PS_MARK(n) do { } while (0)
#endif
// One wave scans the window [ps, pe) of an event (samples at c.samples[base + .], event constants `er`).
//
// Setup (one memory round trip): the ragged head / tail samples, the totals of the <= 64 chunks the window touches (one
// lane each), the digest entries at the window's first and last boundary, one sampled boundary per lane and the first
// rows -- all issued together.  A scan of the chunk totals gives every chunk's offset, kept in the registers of lane c:
// a(t) = E[t] + off[chunk(t)] are the exact sums of [ps, g0 + 8 t) about m.
// Phase 0 (every window): in row r lane L takes block boundary t = 63 r + L (J = g0 + 8t; lane 0 repeats the previous
// row's last boundary so that every block finds its left neighbour one lane below); a row touches at most two chunks,
// whose offsets come from two v_readlane.  The boundary candidate is evaluated, the block (J-8, J) bounded with the
// lane below's left-side values and queued in LDS if it survives (slots from a ballot, no atomics); the queue is
// drained -- interior candidates evaluated from raw samples -- when it may overflow and at the end.  The wave maximum
// decides the common case (no split); top-2 only when something reaches the threshold band.
// Phase 1 (ambiguous windows only, ~1 %): the same sweep with the final maximum known collects the contenders
// (screened gain within 3 delta of the decision level) and the reference's fp64 arithmetic picks among them
// (bs_decide).  A whole-window fp64 scan remains for guard failures and contender overflow.
// AUDIT (diagnostic instance, audit_kernel / ps_audit_bounds only): every row is swept, and every bound the scan forms --
// the corner bound of a block, the two-boundary bound, the group bound -- is compared with the screened gains of the
// candidates it covers, evaluated one by one from the raw samples; counts and smallest margins go to c.dbg.
template <int DT, bool ROWSKIP = true, bool AUDIT = false>
__device__ int scan_window_bs(const DevCfg &c, const EvRef &er, int64_t base, int ps, int pe, int cand_lo, int cand_hi,
                              double thresh, SharedT<64> &sh, unsigned &bad, Work &wk)
{
    constexpr bool WIDE = bs_wide<DT>();
    typedef typename BsTypes<WIDE>::s1_t s1_t;         // first / second moments about m: int32 / fp64 (exact integers), or
    typedef typename BsTypes<WIDE>::s2_t s2_t;         // both int64 with the wide digest
    typedef typename BsTypes<WIDE>::o2_t o2_t;         // second-moment offset of a chunk: fp64 minus 2^52 / int64
    typedef typename BsTypes<WIDE>::Q BsQ_t;
    typedef typename BsTypes<WIDE>::C BsC_t;
    typedef typename BsTypes<WIDE>::E ent_t;           // digest entry: 8 / 16 bytes
    const int lane = threadIdx.x & 63;                 // (one wave; possibly one of several in its workgroup)
    const int n = pe - ps;
    const int g0 = (ps + 7) & ~7, g1 = pe & ~7;
    const int nblk = (g1 - g0) >> 3;
    const long long gb0 = er.boff + (g0 >> 3);
    const int gbl = static_cast<int>(gb0 & (BS_CHUNK - 1));            // chunk of boundary t: (gbl + t) >> 7
    const int nch = ((gbl + nblk) >> BS_CHUNK_LOG) + 1;                // chunks touched
    if (nblk < 4 || nch > BS_MAXCH || c.mode == MODE_EXACT || cand_lo < g0 || cand_hi > g1) {
        // tiny or huge window, or candidates in the ragged head/tail (min_width < 8): exact scan straight from HBM
        wk.exact += 1;                                 // (counters are wave-uniform: scalar registers; lane 0 reports them)
        const long long ex = bs_scan_exact<DT>(bs_cold(c), base + ps, ps, n, cand_lo, cand_hi, thresh, &sh);
        bad |= static_cast<unsigned>(ex >> 32);
        return uni(static_cast<int>(ex));
    }
    PS_STAMP_AT(wk, 7);                                // (diagnostic build) time between scans: recursion control, stack
    PS_MARK(scan_begin);
    const int m = er.m;
    const ent_t *bsw = static_cast<const ent_t *>(c.bsum) + gb0;      // bsw[t]: chunk prefix at boundary t = 0..nblk
    const long long c0 = gb0 >> BS_CHUNK_LOG;
    const int rows = (nblk + BS_STRIDE - 1) / BS_STRIDE;               // nblk + 1 boundaries, 63 new ones per row
    auto row_load = [&](int r) { return bsw[min(max(r, 0) * BS_STRIDE + lane, nblk)]; };
    // everything the window needs before its first boundary, issued together
    const int nh = g0 - ps, nt = pe - g1;              // ragged head [ps, g0) and tail [g1, pe): <= 7 raw samples each
    // (loaded by every lane from a clamped index and converted after all of the setup's loads are out: under a lane
    //  condition the compiler finishes the conversion inside the branch and waits for the sample right there -- two
    //  exposed round trips to HBM in front of the other loads, which is what rounds 1 and 2 did)
    const int ht_idx = lane < 32 ? ps + min(lane, max(nh - 1, 0)) : min(g1 + min(lane - 32, max(nt - 1, 0)), pe - 1);
    const typename Raw<DT>::type ht_raw = static_cast<const typename Raw<DT>::type *>(c.samples)[base + ht_idx];
    // (chunk totals likewise: every lane loads a clamped entry, lanes beyond the window's chunks are masked afterwards)
    const long long cidx = c0 + min(lane, nch - 1);
    int4 ct, cm = make_int4(0, 0, 0, 0);
    if constexpr (WIDE) { ct = c.chunk_tot[2 * cidx]; cm = c.chunk_tot[2 * cidx + 1]; }
    else ct = c.chunk_tot[cidx];
    const ent_t e0 = bsw[0], eN = bsw[nblk];
    // One sampled boundary per lane, spread over the window -- in the instances that skip rows (spine, bridges).  Subtree
    // windows rarely hold a split (95 % of the jobs find nothing): they start at the threshold level without the sampling
    // pass (no gather of 64 scattered entries, ~100 instructions less) and RAISE the level from the gains seen so far
    // whenever the queue has to be drained early (below) -- without that the few windows that do hold a split queue
    // hundreds of blocks (measured: subtree kernel 0.175 -> 0.21 ms).
    // Round 4, narrow digest: the sampling pass is replaced by the COARSE PASS over the group records (below), which
    // every instance runs -- it reads 16 contiguous bytes per 256 samples instead of 64 scattered digest entries, gives
    // the same kind of pruning level, and decides for whole groups of 32 blocks whether the sweep has to look at them.
    constexpr bool GROUPS = !WIDE;
    constexpr bool SAMPLE = WIDE && (ROWSKIP || WIDE || PS_TREE_SAMPLE);      // (wide digest: filtered events, whose subtree windows mostly DO hold splits)
    const int tS = SAMPLE ? min(nblk, lane * rows) : 0;
    ent_t smp = e0;
    if constexpr (SAMPLE) smp = bsw[tS];
    // group records: lane L takes the group boundary c = L, block boundary tC = 32 (G0 + L) - gb0 of the window -- the
    // record of group G0 + L holds the exact sums at that boundary and the amplitudes of the group to its RIGHT
    const long long G0 = (gb0 + BS_GRP - 1) >> BS_GRP_LOG;
    const bool coarse = GROUPS && c.grp != nullptr && c.prune && rows <= 64;                   // (uniform)
    const int ngc = coarse ? min(static_cast<int>(((gb0 + nblk) >> BS_GRP_LOG) - G0), 63) : 0;  // groups entirely inside the window (<= 0: none)
    uint4 gent = make_uint4(0u, 0u, 0u, 0u);
    if constexpr (GROUPS) {
        // (unconditional load from a clamped index, like the rest of the setup: see above)
        const uint4 *gp = coarse ? static_cast<const uint4 *>(c.grp) + G0 : reinterpret_cast<const uint4 *>(bsw);
        gent = gp[coarse ? min(lane, max(ngc, 0)) : 0];
    }
    ent_t ring[BS_D];                                  // rows in flight
    // (a window with a coarse pass sweeps its live rows only, and the rows at its two ends practically always are -- the
    //  first and last groups cannot be bounded: they are requested here, with the rest of the setup, so that the sweep
    //  does not start with a round trip of its own; the other windows start with rows 0 .. 3)
    const bool edge_rows = PS_EDGE_PRELOAD && GROUPS && ngc >= 1;         // (uniform)
    auto pre_row = [&](int i) { return !edge_rows ? i : i == 0 ? 0 : i == 1 ? rows - 1 : i == 2 ? 1 : rows - 2; };
#pragma unroll
    for (int i = 0; i < BS_D; ++i) ring[i] = row_load(pre_row(i));
    if (lane >= nch) ct = make_int4(0, 0, 0, 0);
    const int yab = lane < nch ? (WIDE ? cm.x : ct.y) : 0;
    // head / tail sums (lanes 0..7 head, 32..39 tail)
    const bool ht_in = lane < nh || (lane >= 32 && lane - 32 < nt);
    const int yht = ht_in ? bs_count_of<DT>(c, ht_raw) - m : 0;
    s1_t H1, TL1;
    s2_t H2, TL2;
    if constexpr (WIDE) {
        const long long h1 = oct_sum(static_cast<long long>(yht)), h2 = oct_sum(static_cast<long long>(yht) * static_cast<long long>(yht));
        H1 = lane_get(h1, 7); H2 = lane_get(h2, 7); TL1 = lane_get(h1, 39); TL2 = lane_get(h2, 39);
    } else {
        const int h1 = oct_sum(yht), h2 = oct_sum(yht * yht);      // 7 * 2^28 < 2^31
        H1 = lane_get(h1, 7); H2 = static_cast<double>(lane_get(h2, 7)); TL1 = lane_get(h1, 39); TL2 = static_cast<double>(lane_get(h2, 39));
    }
    // chunk offsets: exclusive scan of the chunk totals over the lanes (rows of 16 suffice for windows up to 15 000 samples)
    const bool many = nch > 16;                        // (uniform)
    s1_t off1;                                         // lane c: sums of [ps, first boundary of chunk c) minus that chunk's prefix base
    o2_t off2;
    s1_t T1;
    s2_t T2;
    int yabs = yab;
    {
#define PS_ROW_STEPS(X) X(0x111, 0xf) X(0x112, 0xf) X(0x114, 0xf) X(0x118, 0xf)
#define PS_BC_STEPS(X) X(0x142, 0xa) X(0x143, 0xc)
#define PS_MSTEP(CTRL, RM) { yabs = max(yabs, dpp_mov<CTRL, RM>(0, yabs)); }
        PS_ROW_STEPS(PS_MSTEP)
        if (many) { PS_BC_STEPS(PS_MSTEP) }
#undef PS_MSTEP
        yabs = lane_get(yabs, many ? 63 : 15);
        if constexpr (WIDE) {
            const long long cs1 = i64_of(ct.x, ct.y), cs2 = i64_of(ct.z, ct.w);
            long long i1 = cs1, i2 = cs2;
#define PS_STEP(CTRL, RM) { const int l1_ = dpp_mov<CTRL, RM>(0, static_cast<int>(i1)), h1_ = dpp_mov<CTRL, RM>(0, static_cast<int>(i1 >> 32)); \
                            const int l2_ = dpp_mov<CTRL, RM>(0, static_cast<int>(i2)), h2_ = dpp_mov<CTRL, RM>(0, static_cast<int>(i2 >> 32)); \
                            i1 += i64_of(l1_, h1_); i2 += i64_of(l2_, h2_); }
            PS_ROW_STEPS(PS_STEP)
            if (many) { PS_BC_STEPS(PS_STEP) }
#undef PS_STEP
            off1 = (i1 - cs1) + (H1 - i64_of(e0.x, e0.y));
            off2 = (i2 - cs2) + (H2 - i64_of(e0.z, e0.w));
            T1 = i64_of(eN.x, eN.y) + lane_get(off1, nch - 1) + TL1;
            T2 = i64_of(eN.z, eN.w) + lane_get(off2, nch - 1) + TL2;
        } else {
            const int cs1 = ct.x;
            const double cs2 = bs_d(i64_of(ct.z, ct.w));                   // < 2^38: exact
            int i1 = cs1;
            double i2 = cs2;
#define PS_STEP(CTRL, RM) { i1 += dpp_mov<CTRL, RM>(0, i1); i2 += dpp_movd<CTRL, RM>(i2); }
            PS_ROW_STEPS(PS_STEP)
            if (many) { PS_BC_STEPS(PS_STEP) }
#undef PS_STEP
            off1 = (i1 - cs1) + (H1 - bs8_s1(e0));
            const double o2 = (i2 - cs2) + (H2 - static_cast<double>(bs8_s2(e0)));   // exact integers below 2^45
            off2 = o2 - BS_BIAS;
            T1 = bs8_s1(eN) + lane_get(off1, nch - 1) + TL1;
            T2 = (bs8_s2_biased(eN) + lane_get(off2, nch - 1)) + TL2;
        }
#undef PS_ROW_STEPS
#undef PS_BC_STEPS
    }
    PS_STAMP_AT(wk, 5);                                // setup loads + head / tail / chunk scans
    PS_MARK(setup_done);
    const double T1d = uni(bs_d(T1)), T2d = uni(bs_d(T2));                // window totals about m
    const double dn = static_cast<double>(n);
    const double Dtot = dn * T2d - T1d * T1d;
    if (!(Dtot > 0.0)) {
        wk.exact += 1;                                 // (counters are wave-uniform: scalar registers; lane 0 reports them)
        const long long ex = bs_scan_exact<DT>(bs_cold(c), base + ps, ps, n, cand_lo, cand_hi, thresh, &sh);
        bad |= static_cast<unsigned>(ex >> 32);
        return uni(static_cast<int>(ex));
    }
    const float nf = static_cast<float>(n);
    const float rn = __builtin_amdgcn_rcpf(nf);
    const float c0f = uni(__builtin_amdgcn_logf(static_cast<float>(Dtot) * rn * rn));
    const float SSt = uni(static_cast<float>(Dtot) * rn);                  // sum of squared deviations of the whole window
    const f2 cc = {c0f, c0f};
    const float dlt = screen_delta_log2(n);
    const float thr_log2 = static_cast<float>(thresh * 1.4426950408889634);
    const float dthr = dlt + 3.0e-6f * nf + 1.0e-6f * fabsf(thr_log2);
    const float LOG2E = 1.4426950408889634f;
    // variance floor: the reference's own fp64 rounding (max|k|^2 * 2^-52 * n; max|k| <= |m| + max|k-m|) and the fp64 D
    // above (kappa_m <= 2^26; wide digest: S2 is rounded to fp64 once, kappa_m <= 2^22 keeps D to 2^-30)
    const float ymaxf = static_cast<float>(yabs);
    const float mabsf = fabsf(static_cast<float>(m)) + ymaxf;
    const float vfloor = uni(fmaxf(mabsf * mabsf * 1.0e-9f, ymaxf * ymaxf * (WIDE ? 2.4e-7f : 1.5e-8f)));
    const unsigned crange = static_cast<unsigned>(cand_hi - cand_lo);
    BsQ_t *queue = reinterpret_cast<BsQ_t *>(sh.q);
    int2 *qblk = reinterpret_cast<int2 *>(queue + BS_QN);             // the queued block's own sums (S1, S2 of its 8 samples; narrow digest)
    BsC_t *cont = reinterpret_cast<BsC_t *>(qblk + BS_QN);
    int4 *ybuf = reinterpret_cast<int4 *>(cont + BS_NC);               // 64 staged blocks: 8 int16 offsets (wide digest: 8 int32)

    int result = -2;
    bool anyflag = false;
    float Tprune, Tc = INFINITY;
    float cbound = INFINITY;                           // bound of the stretch between this lane's sample and the next lane's
    bool hitlike = false;                              // (uniform) a sampled candidate lies above the threshold band
    if constexpr (!SAMPLE) {
        Tprune = (thr_log2 - dthr) - 2.0f * dlt;
    } else {
        // pruning level from the sampled boundary candidates (one per lane, spread evenly over the window: 64 scattered
        // cache lines -- sampling the chunk starts instead, whose sums need no load, was measured: coarser samples skip
        // fewer rows and queue more blocks, spine 0.193 -> 0.225 ms)
        const int J = g0 + 8 * tS;
        const int cs = (gbl + tS) >> BS_CHUNK_LOG;
        const s1_t a1 = bs_a1(smp, static_cast<s1_t>(__shfl(off1, cs)));
        const s2_t a2 = bs_a2(smp, static_cast<o2_t>(__shfl(off2, cs)));
        const BsEval e = bs_eval(bs_d(a1), bs_d(a2), bs_d(static_cast<s1_t>(T1 - a1)), bs_d(static_cast<s2_t>(T2 - a2)),
                                 max(J - ps, 1), max(pe - J, 1), cc, vfloor);
        const bool inr = static_cast<unsigned>(J - cand_lo) <= crange;
        const float bmine = (inr && e.okL && e.okR) ? e.g : -INFINITY;
        float bm = bmine;
        bm = wave_max_f32(bm);
        Tprune = fmaxf(thr_log2 - dthr, bm - 2.0f * dlt) - 2.0f * dlt;
        // A window that holds a split (a sampled gain above the threshold band; rows <= 64: windows up to 32 000
        // samples): the same monotone bound as for an 8-sample block, applied to the whole stretch [J, Jb) up to the
        // NEXT lane's sample (left side from this sample, right side from that one; loose by about Jb - J nats, nothing
        // next to the thousands of nats of a real step).  Nearly every stretch then lies below the pruning level, and
        // the sweep skips the rows that lie in such stretches altogether, loads included: a candidate there is provably
        // more than 2 delta below the winner.  Windows without such a sample sweep every row.
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
    // ---- coarse pass (narrow digest) ---------------------------------------------------------------------------------
    // Lane L evaluates group boundary L (its exact sums: the record's entry plus the chunk offset) like any boundary of
    // the sweep; the largest of those gains sets the pruning level, as the sampled boundaries used to.  Then every group
    // (L-1, L) is bounded: gain(k) <= max(G(L-1) + A(L-1), G(L) + A(L)) for all of its 255 interior candidates, where
    // A = bs_group_slack is what the group's prefix path can gain over its chord (one A per boundary, from the larger
    // amplitudes and the worse mean of its two neighbours).  Rows of the sweep whose blocks all lie in groups below the
    // pruning level are not swept -- nor loaded.  Blocks outside the groups (the window's two ends) are always swept.
    float ghb = INFINITY;                              // bound of group (lane - 1, lane), lanes 1 .. ngc
    if constexpr (GROUPS) {
        if (ngc >= 1) {
            const int tC = static_cast<int>((G0 << BS_GRP_LOG) - gb0) + BS_GRP * lane;
            const bool cin = lane <= ngc;
            const int nlc = cin ? nh + 8 * tC : 1;                          // samples left of the boundary (clamped lanes: any valid value)
            const int cb = min((gbl + (cin ? tC : 0)) >> BS_CHUNK_LOG, nch - 1) << 2;
            const uint2 gentry = make_uint2(gent.x, gent.y);
            const s1_t a1 = bs_a1(gentry, static_cast<s1_t>(bs_from_lane(off1, cb)));
            const s2_t a2 = bs_a2(gentry, static_cast<o2_t>(bs_from_lane(off2, cb)));
            const bool bval = cin && nlc >= 1 && nlc <= n - 1;
            const int nlv = bval ? nlc : 1;
            const double a1d = bs_d(a1), nld = static_cast<double>(nlv);
            const BsEval e = bs_eval(a1d, bs_d(a2), T1d - a1d, T2d - bs_d(a2), nlv, n - nlv, cc, vfloor);
            const bool inr = bval && static_cast<unsigned>(ps + nlc - cand_lo) <= crange;
            const bool eok = bval && e.okL && e.okR;
            float bm = (inr && eok) ? e.g : -INFINITY;
            bm = wave_max_f32(bm);
            Tprune = fmaxf(thr_log2 - dthr, bm - 2.0f * dlt) - 2.0f * dlt;
            // the two neighbours of this boundary: group (L-1, L) -- amplitudes in the record of the lane below -- and (L, L+1)
            const bool hasL = lane >= 1, hasR = lane < ngc;
            const float D1r = __uint_as_float(gent.z), D2r = __uint_as_float(gent.w);
            const float D1l = from_lane_below(D1r), D2l = from_lane_below(D2r);
            const float D1 = fmaxf(hasL ? D1l : 0.0f, hasR ? D1r : 0.0f), D2 = fmaxf(hasL ? D2l : 0.0f, hasR ? D2r : 0.0f);
            // mean of a neighbour minus the mean of the left part, well conditioned: (k S1g / 256 - a1) / k with an exact numerator
            const int a1i = static_cast<int>(a1);
            const int sgl = a1i - from_lane_below(a1i), sgr = from_lane_above(a1i) - a1i;          // S1 of the two groups
            const double ig = 1.0 / static_cast<double>(BS_GRP_SAMPLES);
            const float wLl = static_cast<float>(fma(nld, static_cast<double>(sgl) * ig, -a1d)) * e.r.x;
            const float wLr = static_cast<float>(fma(nld, static_cast<double>(sgr) * ig, -a1d)) * e.r.x;
            const float dmu = static_cast<float>(fma(dn, a1d, -(nld * T1d))) * e.r.x * e.r.y;   // mean left - mean right
            const float A = bs_group_slack(e, D1, D2, hasL ? wLl : wLr, hasR ? wLr : wLl, dmu);
            // (interior candidates have a variance of at least u P / Q on the left, u (n-Q) / (n-P) on the right: none below the floor)
            const float nlf = static_cast<float>(nlv), nrf = nf - nlf;
            const bool okF = e.u.x * nlf >= vfloor * (nlf + static_cast<float>(BS_GRP_SAMPLES)) &&
                             e.u.y * nrf >= vfloor * (nrf + static_cast<float>(BS_GRP_SAMPLES));
            const float GA = (eok && okF) ? e.g + A : INFINITY;             // (A itself is +inf when the expansion does not apply)
            ghb = fmaxf(GA, from_lane_below(GA));
            hitlike = true;
            if constexpr (AUDIT) {
                // lane L walks the 255 candidates inside its group (L-1, L) from the sums at the boundary below
                const int a1b = from_lane_below(a1i);
                const double a2b = __hiloint2double(from_lane_below(__double2hiint(bs_d(a2))), from_lane_below(__double2loint(bs_d(a2))));
                if (lane >= 1 && lane <= ngc) {
                    int x1 = a1b;
                    double x2 = a2b;
                    const int nl0g = nlc - BS_GRP_SAMPLES;
                    float gmx = -INFINITY;
                    for (int j = 1; j < BS_GRP_SAMPLES; ++j) {
                        const int y = bs_count<DT>(c, base + ps + nl0g + j - 1) - m;
                        x1 += y; x2 += static_cast<double>(y) * static_cast<double>(y);
                        const int nlj = nl0g + j;
                        if (nlj >= 1 && nlj <= n - 1) {
                            const BsEval ej = bs_eval(static_cast<double>(x1), x2, T1d - static_cast<double>(x1), T2d - x2, nlj, n - nlj, cc, vfloor);
                            if (ej.okL && ej.okR) gmx = fmaxf(gmx, ej.g);
                        }
                    }
                    bs_audit_note(c.dbg, 4, 5, 8, ghb, gmx, 2.0f * dlt);
                }
                if (lane == 0) atomicAdd(&c.dbg[9], 1ull);
            }
        }
    }
    PS_STAMP_AT(wk, 0);                                // totals, pruning level from the sampled boundaries
    PS_MARK(coarse_done);
#ifdef PS_STAMP
    wk.ph[11] += hitlike;
#endif
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
    const int nl0 = nh + 8 * lane;                     // samples left of this lane's boundary in row 0
    ps_sync<64>();                                     // previous user of sh.q (this wave) is done
    for (int phase = 0; phase < 2; ++phase) {
        Top2 top = {-INFINITY, -INFINITY, -1};
        unsigned flag = 0;
        int qcount = 0;
        // rows to sweep: bit r of `live` (row r covers boundaries 63 r .. 63 r + 63, i.e. the stretches lo .. hi of the
        // lanes' samples; it is skipped when all of them are dead at this phase's pruning level).  Lane r works that out
        // for row r, a ballot makes the mask.  Windows that are not hit-like sweep rows 0 .. rows-1.
        unsigned long long live = ~0ull;
        if constexpr (GROUPS) {
            if (hitlike) {
                // row r holds the blocks 63 r + 1 .. min(63 r + 63, nblk) (block t lies left of boundary t); the group of a block
                // is the group of its left boundary; bit L of `dead`: group (L-1, L) is below the pruning level
                const unsigned long long dead = __ballot(lane >= 1 && lane <= ngc && ghb < Tprune);
                const int tl = BS_STRIDE * lane, th = min(tl + BS_STRIDE, nblk) - 1;
                const int lo = static_cast<int>(((gb0 + tl) >> BS_GRP_LOG) - G0) + 1, hi = static_cast<int>(((gb0 + th) >> BS_GRP_LOG) - G0) + 1;
                const unsigned long long span = ((2ull << (hi - lo)) - 1ull) << max(lo, 0);
                live = __ballot(lane < rows && !(lo >= 1 && hi <= ngc && (dead & span) == span));
                if constexpr (AUDIT) live = __ballot(lane < rows);
            }
        } else if (hitlike) {
            const unsigned long long dead = __ballot(cbound < Tprune);
            const float rr_ = 1.0f / static_cast<float>(rows);
            const int lo = static_cast<int>((static_cast<float>(BS_STRIDE * lane) + 0.5f) * rr_);          // exact for these small integers
            const int hi = min(63, static_cast<int>((static_cast<float>(BS_STRIDE * lane + BS_STRIDE) + 0.5f) * rr_));
            const unsigned long long span = (hi - lo >= 63) ? ~0ull : (((1ull << (hi - lo + 1)) - 1ull) << lo);
            live = __ballot(lane < rows && (dead & span) != span);
        }
#if PS_ONE_ROW_LOOP
        // Windows that sweep every row (no coarse pass, no hit-like sample) walk the same loop as the others: their mask is
        // "all rows", 64 at a time (one copy of the row code instead of two: the scan kernels are 40 KB each, several of them
        // run on a pair of CUs at once, and the instruction cache of the pair holds 64 KB).
        int rbase = 0;                                 // (uniform) first row of the mask
        if (!hitlike) live = rows >= 64 ? ~0ull : (1ull << rows) - 1ull;
        auto take_row = [&]() {                        // next live row, -1: none left (uniform)
            if (live == 0ull) {
                if (hitlike || rbase + 64 >= rows) return -1;
                rbase += 64;
                live = rows - rbase >= 64 ? ~0ull : (1ull << (rows - rbase)) - 1ull;
            }
            const int r = __builtin_ctzll(live);
            live &= live - 1ull;
            return rbase + r;
        };
#else
        auto take_row = [&]() {                        // next live row, -1: none left (uniform)
            if (live == 0ull) return -1;
            const int r = __builtin_ctzll(live);
            live &= live - 1ull;
            return r;
        };
#endif
        auto drain = [&]() {
            // drain: interior candidates of the queued blocks
            PS_MARK(drain_begin);
            ps_sync<64>();
            PS_STAMP_AT(wk, 1);                        // boundary sweep
            if constexpr (!WIDE && PS_DRAIN_FILTER) {
                // First the queued blocks are judged again, from both of their boundaries (bs_block_bound2; one block per
                // lane).  The sweep's corner bound keeps 3-5 % of the blocks of a window without a split; this one keeps
                // practically none of them, and the window then needs neither the blocks' samples -- a dependent round trip
                // to HBM -- nor the evaluation of their 7 x 23 candidates.  Blocks on the slope of a step stay, as before.
                int kept = 0;
                for (int r = 0; r < qcount; r += 64) {
                    const int idx = min(r + lane, qcount - 1);
                    const BsQ_t q = queue[idx];
                    const int2 bl = qblk[idx];
                    const int nlq = bs_q_j(q, ps) - ps, nlp = nlq - 8;
                    const int a1q = bs_q_a1(q), a1p = a1q - bl.x;
                    const double a2q = q.a2, a2p = a2q - static_cast<double>(static_cast<unsigned>(bl.y));
                    bool stay = r + lane < qcount;
                    if (nlp >= 1) {
                        const BsEval eq = bs_eval(bs_d(a1q), a2q, bs_d(T1 - a1q), T2 - a2q, nlq, n - nlq, cc, vfloor);
                        const BsEval ep = bs_eval(bs_d(a1p), a2p, bs_d(T1 - a1p), T2 - a2p, nlp, n - nlp, cc, vfloor);
                        stay = stay && !(bs_block_bound2(ep, eq, nlp, n, SSt, rn) < Tprune);
                    }
                    const unsigned long long sm = __ballot(stay);
                    ps_sync<64>();                       // every lane has read its entry
                    if (stay) {
                        const int slot = kept + lanes_below(sm);
                        queue[slot] = q;
                        qblk[slot] = bl;
                    }
                    kept += __popcll(sm);
                    ps_sync<64>();
                }
                qcount = kept;
            }
#ifdef PS_STAMP
            if (qcount) { wk.ph[9] += 1; wk.ph[10] += qcount; }           // (diagnostic build) non-empty drains, blocks drained
#endif
            for (int r = 0; r < qcount; r += 64) {
                // (a) one queued block per lane: its 8 samples, as int16 offsets from m, go to LDS
                const int nb = min(64, qcount - r);
                if (lane < nb) {
                    const int64_t gq = base + bs_q_j(queue[r + lane], ps) - 8;
                    int y[8];
#pragma unroll
                    for (int w = 0; w < 8; ++w) y[w] = bs_count<DT>(c, gq + w) - m;
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
            PS_MARK(drain_end);
            PS_STAMP_AT(wk, 2);                        // drain
        };
        // early drain (the queue is filling: this window probably holds a split): afterwards the pruning level follows the
        // best gain seen so far -- a block whose bound lies more than 2 delta below a gain that WAS reached cannot hold the
        // winner nor a contender (the same argument as for the sampled level, with real gains instead of sampled ones)
        auto drain_early = [&]() {
            drain();
            if (phase == 0) {
                float bx = top.b;
                bx = wave_max_f32(bx);
                Tprune = fmaxf(Tprune, fmaxf(thr_log2 - dthr, bx - 2.0f * dlt) - 2.0f * dlt);
            }
        };
        // row r: lane L takes boundary t = 63 r + L (J = g0 + 8 t); the row's boundaries lie in at most two chunks
        struct RowOut { s1_t a1; s2_t a2; int nl; float ge, hb; bool blk, prunable, unsure; };
        auto row_eval = [&](int r, const ent_t &cur_in) -> RowOut {
            PS_MARK(row_eval_begin);
            ent_t cur = cur_in;
#ifdef PS_STAMP
            wk.ph[8] += 1;                                                 // (diagnostic build) rows swept
#endif
            const bool first_row = r == 0;
            const int tb = gbl + BS_STRIDE * r;                            // (uniform) chunk-relative index of the row's first boundary
#if PS_ROW_OFFSETS_BPERMUTE
            // every lane fetches the offsets of ITS chunk from the lane that holds them (ds_bpermute: three or four LDS-pipe
            // instructions and the index) -- two v_readlane per value plus moves and selects were 22 instructions a row
            const int cb = min((tb + lane) >> BS_CHUNK_LOG, nch - 1) << 2;
            const s1_t o1 = bs_from_lane(off1, cb);
            const o2_t o2 = bs_from_lane(off2, cb);
#else
            const int cA = min(tb >> BS_CHUNK_LOG, nch - 1), cB = min(cA + 1, nch - 1);
            const int lsw = ((tb >> BS_CHUNK_LOG) + 1) * BS_CHUNK - tb;    // first lane in the second chunk (>= 64: none)
            const bool second = lane >= lsw;
            const s1_t o1A = lane_get(off1, cA), o1B = lane_get(off1, cB);   // (both read unconditionally: a select, not a branch)
            const o2_t o2A = lane_get(off2, cA), o2B = lane_get(off2, cB);
            const s1_t o1 = second ? o1B : o1A;
            const o2_t o2 = second ? o2B : o2A;
#endif
            const int nl = nl0 + 8 * BS_STRIDE * r, J = ps + nl;
            const float nlf = static_cast<float>(nl);
            const double nld = static_cast<double>(nl);
            const s1_t a1 = bs_a1(cur, o1);
            const s2_t a2 = bs_a2(cur, o2);
            // screened gain of the boundary (bs_eval, with the running nl and the right side from the totals)
            const double a1d = bs_d(a1), b1d = T1d - a1d;                  // (exact: |S1| < 2^53)
            double a2d, b2d;
            if constexpr (WIDE) { a2d = d_of_i64(a2); b2d = d_of_i64(T2 - a2); }
            else { a2d = a2; b2d = T2d - a2; }
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
            // (bitwise, not short-circuit: with && the compiler puts the bound under an exec-mask branch)
            const bool prunable = static_cast<bool>(static_cast<int>(pokL) & static_cast<int>(okR) & static_cast<int>(nl >= 9));
            return RowOut{a1, a2, nl, ge, fmaxf(h0, h1), blk, prunable, inr && !(okL && okR)};
        };
        // (the part with side effects, in row order: per-lane top-2, the block queue, the contender list)
        auto row_commit = [&](const RowOut &o) {
            PS_MARK(row_commit_begin);
            if constexpr (AUDIT && !WIDE) {
                if (phase == 0) {
                    // the 7 candidates inside the block (J - 8, J), from the sums at the boundary below and the raw samples
                    const int a1p = from_lane_below(o.a1);
                    const double a2p = __hiloint2double(from_lane_below(__double2hiint(o.a2)), from_lane_below(__double2loint(o.a2)));
                    const int nlp = o.nl - 8;
                    if (o.blk && nlp >= 1 && o.nl <= n - 1) {
                        int x1 = a1p;
                        double x2 = a2p;
                        float gmx = -INFINITY;
                        for (int u = 1; u < 8; ++u) {
                            const int y = bs_count<DT>(c, base + ps + nlp + u - 1) - m;
                            x1 += y; x2 += static_cast<double>(y) * static_cast<double>(y);
                            const BsEval ej = bs_eval(static_cast<double>(x1), x2, T1d - static_cast<double>(x1), T2d - x2, nlp + u, n - nlp - u, cc, vfloor);
                            if (ej.okL && ej.okR) gmx = fmaxf(gmx, ej.g);
                        }
                        if (o.prunable) bs_audit_note(c.dbg, 0, 1, 6, o.hb, gmx, 2.0f * dlt);
                        const BsEval eq = bs_eval(bs_d(o.a1), o.a2, bs_d(T1 - o.a1), T2 - o.a2, o.nl, n - o.nl, cc, vfloor);
                        const BsEval ep = bs_eval(bs_d(a1p), a2p, bs_d(T1 - a1p), T2 - a2p, nlp, n - nlp, cc, vfloor);
                        bs_audit_note(c.dbg, 2, 3, 7, bs_block_bound2(ep, eq, nlp, n, SSt, rn), gmx, 2.0f * dlt);
                    }
                }
            }
            top2_push(top, o.ge, o.nl);
            flag |= static_cast<unsigned>(o.unsure);
            // A row with a boundary gain that lies above the pruning level by more than the level's own margin: the level
            // follows that gain BEFORE the row's blocks are judged (the argument of drain_early: a block bounded more than
            // 2 delta below a gain that was reached holds neither the winner nor a contender).  Without it a window that
            // holds a step queues every block of the slope that leads up to it -- each row lies above everything seen
            // before -- and 5 % of the subtree windows did 80 % of the kernel's block drains.  Windows without a step
            // never get here (two instructions per row).
            if (phase == 0 && __ballot(o.ge - 4.0f * dlt > Tprune) != 0ull) {
                float bx = o.ge;
                bx = wave_max_f32(bx);
                Tprune = fmaxf(Tprune, fmaxf(thr_log2 - dthr, bx - 2.0f * dlt) - 2.0f * dlt);
            }
            const bool keep = static_cast<bool>(static_cast<int>(o.blk) & static_cast<int>(!(o.prunable && o.hb < Tprune)));
            const unsigned long long km = __ballot(keep);
#ifdef PS_STAMP
            wk.ph[6] += __popcll(__ballot(keep && !o.prunable));               // (diagnostic build) kept because not prunable
#endif
            if (km) {
                int2 bsum = make_int2(0, 0);
                if constexpr (!WIDE && PS_DRAIN_FILTER) {
                    // the block's own sums: this boundary's minus the one below (all lanes are here: the branch is the wave's)
                    const double a2b = __hiloint2double(from_lane_below(__double2hiint(o.a2)), from_lane_below(__double2loint(o.a2)));
                    bsum = make_int2(o.a1 - from_lane_below(o.a1), static_cast<int>(static_cast<unsigned>(o.a2 - a2b)));
                }
                if (keep) {
                    BsQ_t q;
                    bs_q_put(q, ps + o.nl, ps, o.a1, o.a2);
                    const int slot = qcount + lanes_below(km);
                    queue[slot] = q;
                    if constexpr (!WIDE && PS_DRAIN_FILTER) qblk[slot] = bsum;
                }
                qcount += __popcll(km);
            }
            if (phase) PS_COLLECT(o.ge >= Tc, o.ge, ps + o.nl, o.a1, o.a2)
            PS_MARK(row_commit_end);
        };
        auto do_row = [&](int r, const ent_t &cur) { row_commit(row_eval(r, cur)); };
        // One row at a time (no interleaving of rows: the four waves of the SIMD cover each other's latencies, and a row
        // evaluated alone keeps the kernel at 128 registers); its slot of the ring is refilled as soon as it is free.
        if (!PS_ONE_ROW_LOOP && !hitlike) {
            if (phase) {
#pragma unroll
                for (int i = 0; i < BS_D; ++i) ring[i] = row_load(i);
            }
            for (int r0 = 0; r0 < rows; r0 += BS_D) {
#pragma unroll
                for (int i = 0; i < BS_D; ++i) {
                    if (r0 + i < rows) {
                        if (qcount > BS_EARLY) drain_early();
                        do_row(r0 + i, ring[i]);
                    }
                    // (unconditional, clamped past the end: a load under a condition makes the compiler's count of the loads
                    //  in flight imprecise, and every row then waits for the newest request instead of its own)
                    ring[i] = row_load(r0 + i + BS_D);
                }
            }
        } else {
            // live rows only (a window that holds a split: typically 2 .. 4 of 20)
            int rr[BS_D];                              // (uniform) row of every slot, -1: none
#pragma unroll
            for (int i = 0; i < BS_D; ++i) {
                // (phase 0 of a window with a coarse pass: the slot already holds a row of the window's ends; taken if live)
                const int pr = pre_row(i);
                // (one loop for all windows: the rows requested with the setup are 0 .. BS_D - 1 when the window has no coarse pass)
                const bool have = (PS_ONE_ROW_LOOP ? (hitlike ? GROUPS && edge_rows : true) : GROUPS && edge_rows) && phase == 0 &&
                                  pr >= 0 && pr < min(rows, 64) && ((live >> pr) & 1ull) != 0ull;
                if (have) { rr[i] = pr; live &= ~(1ull << pr); }
                else { rr[i] = take_row(); if (!PS_RING_GUARD || rr[i] >= 0) ring[i] = row_load(rr[i]); }
            }
            for (bool any = true; any;) {
                any = false;
#pragma unroll
                for (int i = 0; i < BS_D; ++i) {
                    if (rr[i] >= 0) {
                        if (qcount > BS_EARLY) drain_early();
                        do_row(rr[i], ring[i]);
                        any = true;
                    }
                    if (any) { rr[i] = take_row(); if (!PS_RING_GUARD || rr[i] >= 0) ring[i] = row_load(rr[i]); }
                }
            }
        }
        drain();
        if (phase == 1) break;
        // the wave maximum decides the common case; top-2 (DPP) only when something reaches the threshold band
        anyflag = __ballot(flag != 0) != 0ull;
        float mx = top.b;
        const float ab = wave_max_f32(mx);
        float as = -INFINITY;
        int ai = -1;
        if (!anyflag && ab < thr_log2 - dthr) {
            result = -1;
        } else if (!anyflag) {
#define PS_STEP(CTRL, RM) { const float ob = dpp_movf<CTRL, RM>(-INFINITY, top.b), os = dpp_movf<CTRL, RM>(-INFINITY, top.s); \
                            const int oi = dpp_mov<CTRL, RM>(-1, top.i); top2_merge(top, ob, os, oi); }
            PS_DPP_STEPS(PS_STEP)
#undef PS_STEP
            as = __int_as_float(lane_get(__float_as_int(top.s), 63));
            ai = lane_get(top.i, 63);
            if (ab > thr_log2 + dthr && as < ab - 2.0f * dlt) result = ps + ai;
        }
#ifdef PS_NOEXACT
        if (result == -2) result = ab < thr_log2 ? -1 : ps + ai;       // timing experiment only: never the product
#endif
        PS_MARK(decide_done);
        PS_STAMP_AT(wk, 3);                            // wave maximum / top-2, decision
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
    const bool sums_exact = WIDE || static_cast<double>(n) * static_cast<double>(mabsf) * static_cast<double>(mabsf) < 9007199254740992.0;
    if (result == -2 && !anyflag && ccount <= BS_NC && sums_exact) {
        ps_sync<64>();                              // contender stores visible to the other lanes
        const long long dr = bs_decide<DT>(bs_cold(c), m, cont, ccount, ps, n, T1, T2, thresh);
        result = uni(static_cast<int>(dr));
        wk.exact += 1; wk.near += uni(static_cast<int>(dr >> 32));
    }
    PS_MARK(scan_end);
    PS_STAMP_AT(wk, 4);                                // contenders + fp64 decision
    if (c.mode == MODE_VERIFY || result == -2) {
        wk.exact += (1LL << 32);                       // high word: full exact scans
        const long long exr = bs_scan_exact<DT>(bs_cold(c), base + ps, ps, n, cand_lo, cand_hi, thresh, &sh);
        bad |= static_cast<unsigned>(exr >> 32);
        const int ex = uni(static_cast<int>(exr));
        if (result != -2 && ex != result) {
            bad |= ST_VERIFY_MISMATCH;
            if (wk.dbg[1] == 0) { wk.dbg[0] = ps; wk.dbg[1] = pe; wk.dbg[2] = result; wk.dbg[3] = ex; }
        }
        return ex;
    }
    return result;
}

// The window [ps, pe) of an event whose digest is aligned to the TRACE (EvRef::ph = (event start) mod 8 != 0): the same scan in the
// coordinates of the 8-aligned stretch that starts ph samples before the event -- every position shifted by ph, the sample base
// by -ph, the event's first block er.boff = floor(start / 8).  All of scan_window_bs is relative arithmetic (n = pe - ps, n_l =
// J - ps, candidates, the ragged head and tail), so nothing else changes; ph = 0 is the identity.
template <int DT, bool ROWSKIP = true, bool AUDIT = false>
__device__ __forceinline__ int scan_window_ph(const DevCfg &c, const EvRef &er, int64_t base, int ps, int pe, int cand_lo, int cand_hi,
                                              double thresh, SharedT<64> &sh, unsigned &bad, Work &wk)
{
    const int ph = er.ph;
    const int r = scan_window_bs<DT, ROWSKIP, AUDIT>(c, er, base - ph, ps + ph, pe + ph, cand_lo + ph, cand_hi + ph, thresh, sh, bad, wk);
    return r >= 0 ? r - ph : r;
}

// ---- K2 from the K0 digest: per-segment statistics without a second pass over the samples ------------------
// Segment.mean/std/min/max (core.py:209-223).  One wave per segment (workgroups stride over the segments):
// S1, S2 of the full blocks inside the segment from the chunk prefix and the chunk totals, min/max from the per-block
// table, the ragged ends (<= 7 samples each) from the samples.  mean = (m + S1/n) q, std = sqrt(S2/n - (S1/n)^2) q
// (population; formed about m, so nothing cancels), min/max exact.  n_seg = bounds_off[n_ev] + n_ev is read on the
// device; `stats_cap` bounds the writes.  (Narrow digest only: the wide one keeps no min/max table.)
template <int DT>
__global__ __launch_bounds__(64) void segstat_bs_kernel(DevCfg c, const int64_t *ev_start, const int64_t *ev_len, int32_t n_ev,
                                                        const int32_t *bounds, const int64_t *bounds_off, ps_segstat *stats,
                                                        int64_t stats_cap, unsigned *status, const AsmHeader *hdr)
{
    const int lane = threadIdx.x;
    // a failed stitch or a refused digest leaves no valid boundaries: the host redoes the call on another path
    if ((hdr && hdr->fail) || (*status & ~ST_VERIFY_MISMATCH) != 0u) return;
    const int64_t n_seg = min(bounds_off[n_ev] + n_ev, stats_cap);
    const uint2 *bs = static_cast<const uint2 *>(c.bsum);
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
        int a = sidx == 0 ? 0 : bounds[boff + sidx - 1];
        int b = sidx == cnt ? static_cast<int>(ev_len[e]) : bounds[boff + sidx];
        int64_t base = ev_start[e];
        const int4 info = c.ev_info[e];
        const int m = info.x;
        const long long eb = (static_cast<long long>(static_cast<unsigned>(info.w)) << 32) | static_cast<unsigned>(info.z);
        const int n = b - a;
        if (a < 0 || b < a || b > ev_len[e]) continue;                     // (never with valid boundaries)
        // (a trace-aligned digest: the event starts info.y samples into its first block -- shifted coordinates, see scan_window_ph)
        a += info.y; b += info.y; base -= info.y;
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
            const long long c0 = gb0 >> BS_CHUNK_LOG, c1 = gb1 >> BS_CHUNK_LOG;
            for (long long cc = c0 + lane; cc < c1; cc += 64) {
                const int4 t = c.chunk_tot[cc];
                s1 += static_cast<double>(t.x); s2 += bs_d(i64_of(t.z, t.w));
            }
            for (long long gb = gb0 + lane; gb < gb1; gb += 64) {
                const int w = c.blk_mm[gb];
                mn = min(mn, static_cast<int>(static_cast<short>(w & 0xffff))); mx = max(mx, w >> 16);
            }
            if (lane == 0) {
                const uint2 p0 = bs[gb0], p1 = bs[gb1];
                s1 += static_cast<double>(bs8_s1(p1)) - static_cast<double>(bs8_s1(p0));
                s2 += static_cast<double>(bs8_s2(p1)) - static_cast<double>(bs8_s2(p0));
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
