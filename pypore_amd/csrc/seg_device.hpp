// seg_device.hpp -- device side of libporeseg: window scan, recursion drivers, kernels.
//
// gfx950 only (wave64, 256 CUs, 160 KB LDS/CU).  One workgroup of NT threads works on one
// job; all control flow of the recursion is workgroup-uniform (decisions are broadcast through
// LDS), the per-candidate work is spread over the lanes.
//
// Reference functions restated here (PyPore/cparsers.pyx):
//   var_c                 :31-38    -> ref_var()
//   _best_split_stepwise  :157-178  -> scan_window()  (screen in fp32, decide exactly in fp64)
//   _recursive_split      :180-203  -> find_split() + spine_kernel / tree_kernel
//   _best_single_split    :134-155  -> single_scan_kernel (mode 1)
//   _best_split_stepwise_score :222-249 -> single_scan_kernel (mode 0)
//
// How a window [ps,pe) is scanned (DESIGN.md "window scan"):
//   1. stage   coalesced HBM loads -> integer ADC counts in LDS; min/max of the window
//   2. prefix  per-thread chunk sums of (k-m0), (k-m0)^2 in int32/uint32, workgroup exclusive
//              scan in fp64 (the sums are integers, so every order gives the same bits)
//   3. screen  every candidate's gain in fp32 from exact integer moment sums that are
//              re-centred per thread chunk (no cancellation); error <= delta (proved bound)
//   4. decide  workgroup top-2 reduction; if the best candidate is unique by more than
//              2*delta and clear of the threshold by more than delta the answer is final;
//              otherwise (rare) every candidate is re-evaluated in fp64 with the reference's
//              exact operation order and first-maximum tie-break.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "poreseg.h"

namespace ps {

constexpr int LDS_STACK = 128;   // DFS stack entries kept in LDS before spilling to HBM
constexpr int MAX_WAVES = 16;
constexpr int OBUF = 256;          // (>= BR_MAX: a bridge buffers its whole chain in LDS)
constexpr int BR_MAX = 256;       // bridge chain: anchors a seam may add before it must have joined (densely stepped
                                  // data: two chains pick the best of ~10 steps per window and can take dozens of
                                  // anchors to meet; a seam that gives up sends the whole call to the host stitch)
constexpr int LST_MAX = 256;
// Round 5: seams that the bridges gave up on (BR_MAX anchors without meeting a downstream list: densely stepped data, where two
// chains can stay out of step for a long stretch) and open tiles that the true chain enters exactly at their start get a second
// chance on the device before the call falls back to the host stitch: the look-ahead kernel continues them with room for
// EXT_MAX more anchors each, in a side buffer (anchor i >= BR_MAX of seam g lives at ext[ext_slot[g] * EXT_MAX + i - BR_MAX]).
// Round 5, long stretches without splits.  The windows of a chain between two anchors lie on a lattice a + J * W/2, and whether
// window J holds a split does not depend on who asks.  In the look-ahead kernel an OWNER (the workgroup that walks a seam) that
// has scanned LAT_W windows of one find_split without a hit LISTS the stretch: lattice origin, a fresh tag, the chunk (LAT_W
// windows) it is in.  Workgroups that are through with their own seams HELP: they scan chunks ahead of the listed owners -- a
// few at first, more as the owner gets further -- and publish each chunk's first hit as one 64-bit word under the stretch's
// tag.  An owner at a chunk boundary takes a published result instead of scanning the chunk, and scans it itself when there is
// none: nobody waits for anybody's result.  Forced splits (start + max_width) keep the lattice when max_width is a multiple of
// W/2; a real split starts a new one (listed again once it has gone LAT_W windows).  Helpers stay only if the single-wave bridge
// kernel has seen a deferred seam with an open tile ahead (ctl[0]: never on densely stepped data), and leave when nothing is
// listed and every workgroup is through with its own seams -- or nothing has been listed for ~0.1 ms.
constexpr int LAT_D = 256;         // listed stretches' slots (one per seam that ever lists)
constexpr int LAT_W = 16;          // windows per chunk
constexpr int LAT_C = 2048;        // chunks per lattice (32 768 windows: 1.6e8 samples at W = 10 000)
constexpr int LAT_TAGS = 4096;     // listings per call (tags are unique per context until the host clears the results: 24 bits)
struct LatHelp {
    unsigned long long *ctl;       // [0] hint (bridge_kernel) [1] slots taken [2] tags taken [3] workgroups through with their own seams
                                   // [4] chunk results published by helpers [5] published chunks an owner took instead of scanning (counters)
    unsigned long long *state;     // [LAT_D] lattice origin | tag << 32 | first helped chunk << 56 (one word: read and written whole)
    int *seam;                     // [LAT_D] the seam (tile) of the slot
    int *prog;                     // [LAT_D] 0: nothing to help with; else 1 + the chunk the owner is in (zeroed with the call's status block)
    unsigned long long *res;       // [LAT_D * LAT_C] tag << 40 | offset of the chunk's first hit << 32 | its split (0xffffffff: none)
    unsigned tag_base;             // first tag of this call
    int stay;                      // 1 (option lat_help = 2): helpers do not leave on idle polls, only when every workgroup is through (tests)
};
constexpr LatHelp LAT_NONE = {nullptr, nullptr, nullptr, nullptr, nullptr, 0u, 0};
__device__ __forceinline__ unsigned long long lat_ld(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int lat_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lat_st(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lat_st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
constexpr int EXT_MAX = 16384;     // further anchors of an extended seam when few seams need them (<= EXT_SLOTS) ...
constexpr int EXT_SLOTS = 64;
constexpr int EXT_MAX_MANY = 2048; // ... and when many do (a batch of events with a failed seam each): the side buffer holds
constexpr int EXT_ROUNDS = 4;      // EXT_SLOTS * EXT_MAX anchors either way; the stride is fixed by the first round of a call
constexpr int GS_LOG = 8;          // gather_scan_kernel: items per workgroup (256); tree jobs add their counts to blk[job >> GS_LOG]
constexpr int QMAX = 256;          // blocks of candidates queued for full evaluation per window
constexpr int PBLK = 8;            // candidates per pruning block      // anchors of the downstream tile cached in LDS for membership tests        // buffered outputs per job (int2 anchors / 2x int boundaries)

enum : int { KIND_NONE = 0, KIND_HIT = 1, KIND_EARLY = 2, KIND_LATE = 3, KIND_STOP = 4 };   // STOP: gave up at stop_lim (spine tiles)
enum : unsigned { ST_OFF_GRID = 1u, ST_OUT_OVERFLOW = 2u, ST_STACK_OVERFLOW = 4u, ST_VERIFY_MISMATCH = 8u };
enum : int { MODE_FAST = 0, MODE_EXACT = 1, MODE_VERIFY = 2 };

struct DevCfg {
    const void *samples;
    int dtype, off_counts;
    double dc_counts;       // fp32 samples only: the level, in counts of `quantum`, that the caller subtracted upstream (ps_requantise's centre):
                            // never added to a sample -- the gains are shift invariant --, used to judge near ties against the reference's own
                            // rounding noise, whose cumsums run on the uncentred values (seg_bs.hpp: bs_decide)
    float inv_q;
    float noise_k;          // near-tie accounting of the wide route: margins within noise_k x (the two sides' noise estimates) count (bs_decide)
    double q, q2;
    int mw, maxw, W, half;
    double min_gain;
    int mode;               // MODE_*
    int prune;              // 1: block-bound pruning in the screen (default), 0: evaluate every candidate
    int lds_cap;            // samples that fit the dynamic LDS window buffer
    const void *bsum;       // per 8-sample block: chunk-exclusive prefix of (sum (k-m), sum (k-m)^2), all events -- 8 bytes (25 + 39
                            // bits), wide digest 16 bytes (two int64), seg_bs.hpp; nullptr: LDS-window scan
    const int4 *ev_info;    // per event: (m, -, block offset lo, block offset hi)
    const int4 *chunk_tot;  // per chunk of 128 blocks: (S1, max |k-m|, S2 as int64); wide digest: two entries (S1, S2 as int64), (max |k-m|, -, -, -)
    int *blk_mm;            // per block: min / max of k-m as two int16 (written by K0 when statistics are wanted) or nullptr
    // single-pass file route (ps_detect_segment_trace): K0 over the whole trace also judges every block against the detector's
    // threshold -- 2 bits per block (0 all at or above, 1 all below, 2 mixed), one byte per lane -- and leaves min / max of k-m
    // per 128-block chunk; the edge kernel reads those 1/64 of the samples' bytes instead of the samples (nullptr: not asked for)
    unsigned char *blk_cls;
    int2 *cls_mm;
    int cls_kthr;           // below(k) <=> k < cls_kthr (counts; the host derives it from the detector's own predicate, below_thr)
    const void *grp;        // per group of 32 blocks (256 samples; narrow digest): the digest entry of its first block + (D1, D2) as fp32,
                            // the bridge amplitudes of the group bound (seg_bs.hpp); nullptr: no coarse pass
    int bs_wide;            // (host side) the digest is the 64-bit one: kernels compiled for DT | DT_WIDE
    int k0_unaligned;       // 1: K0's fast route may load 16 bytes from sample-aligned addresses (probed at ps_create), 0: 16-byte-aligned only
    const double *pre_c;    // exact route for float64 input on no grid (ps_segment_exact_f64): the reference's own prefix sums, c = cumsum(x) and
    const double *pre_c2;   // c2 = cumsum(x * x) per event, strictly sequential like numpy's (cparsers.pyx:110-111); nullptr on every other route
    unsigned long long *dbg;  // diagnostics scratch (12 words) or nullptr
    int rep_eval, rep_stage, rep_sum;   // diagnostics: repeat a phase to measure its marginal cost (normally 1)
};

// What a block-sum window scan needs of its event: the centre m of the digest's sums, the global index of the block that holds the
// event's first sample, and -- round 6 -- the PHASE ph = (event start) mod 8 when the digest's blocks are aligned to the TRACE, not to
// the event (ps_detect_segment_trace: one K0 pass over a whole file trace serves the detector and every event cut out of it).
// ph = 0 on every other route.  A window [ps, pe) of such an event is the window [ps + ph, pe + ph) of the 8-aligned stretch that
// starts ph samples earlier: scan_window_ph runs the scan unchanged in those shifted coordinates.
struct EvRef { int m; long long boff; int ph; };

struct SpineJob {           // speculative spine of one tile: rec(start, end) without left subtrees
    int64_t base;           // offset of the event in the sample array
    int32_t start, end;     // chain anchor, event length
    int32_t stop;           // stop after the first spine anchor >= stop
    int32_t out_cap;
    int64_t out_off;        // into the anchor list storage (int2 entries)
    int32_t first_tile;     // global index of the event's first tile
    int32_t ntiles;         // tiles of this event
    int32_t tile_len;       // tile t of the event starts at t * tile_len
    int32_t ev;             // event index
    int64_t vbase;          // sum of the lengths of the preceding events (layout of the tree output regions)
};

struct TreeJob {            // full in-order traversal of rec(start, end), first window index j0
    int64_t base;
    int32_t start, end, j0;
    int32_t out_cap;
    int64_t out_off;        // into the private boundary scratch (int32) and the spill stack (int2)
    int32_t m, pad_;        // the event's centre (first count), pad_ = its phase (EvRef::ph), and first block (K0 digest; 0 on the LDS-window path)
    int64_t boff;
};

typedef short lds_t;               // window samples in LDS: ADC counts as int16 (windows that do not fit go the exact HBM path)

struct QEnt { int j, jend, p1, r1; unsigned p2, r2; int cL, cR; };

// Per-workgroup scratch in LDS.  The 64-thread (single wave) variant used by the block-sum scan
// keeps it small so that 16 workgroups fit a CU.
template <int NT> struct SharedT {
    static constexpr int NWV = NT / 64;
    static constexpr int OB = NT == 64 ? BR_MAX : OBUF;      // buffered outputs (a bridge buffers its whole chain)
    static constexpr int QN = NT == 64 ? 208 : QMAX;         // queued blocks (NT == 64: the block-sum scan's LDS, seg_bs.hpp: 6 656 bytes)
    static constexpr int LN = NT == 64 ? 128 : LST_MAX;      // cached downstream anchors
    static constexpr int SN = NT == 64 ? 64 : LDS_STACK;     // DFS stack entries before spilling
    double wsum1[NWV], wsum2[NWV];
    double wbest[NWV];
    int widx[NWV];
    float fbest[NWV], fsecond[NWV];
    int fidx[NWV];
    int wmin[NWV], wmax[NWV];
    unsigned wflag[NWV];
    int bcast;
    int2 pop;
    int2 stack[SN];
    int lst[LN];
    float wmaxf[NWV];
    int qn;
    QEnt q[QN];
    int2 obuf[OB];          // results are buffered here and written to HBM once per job: a global
                            // store issued between scans would sit in front of the next window's
                            // staging loads (vmcnt retires in order) and stall the whole workgroup
};
typedef SharedT<1024> Shared;  // (size reference for the host-side LDS budget: the largest variant)

#ifndef PS_BS_MINW
#define PS_BS_MINW 4          // waves per SIMD the single-wave (block-sum) kernels are compiled for
#endif
// Registers of the block-sum scan kernels.  Round 3: 128 (four waves per SIMD; the scan body of seg_bs.hpp is written for
// it: small row groups, chunk offsets in lane registers, cold fp64 paths out of line) -- a window spends a third of its
// time in dependent memory round trips (setup, drain, decision), which two waves per SIMD could not cover.  Round 2 ran
// them at 232 (two waves plus room for a wave of another call's K0).  The attribute counts architectural VGPRs, half of
// the unified file on this part: 64 -> 128.  A kernel whose launch bounds allow less -- the NT >= 256 instances, 128 --
// ignores it.
#ifndef PS_SCAN_VGPRS
#define PS_SCAN_VGPRS 64
#endif
#define PS_SCAN_REGS __attribute__((amdgpu_num_vgpr(PS_SCAN_VGPRS)))
// The seam bridges are a few hundred windows in chains of two or three: latency is all that matters to them, occupancy
// nothing -- they are compiled for two waves per SIMD and keep everything in registers (at 128 registers: 256 bytes of
// scratch per lane, 0.056 -> 0.071 ms).
#ifndef PS_BRIDGE_MINW
#define PS_BRIDGE_MINW 2
#endif
#define PS_BRIDGE_REGS __attribute__((amdgpu_num_vgpr(256 / PS_BRIDGE_MINW - 12)))
struct Work {
    long long windows, cands, exact;
    long long near;          // contender decisions whose margin lies inside the device-vs-glibc logarithm noise (seg_bs.hpp: bs_decide)
    long long dbg[4];        // verify mode: first disagreement (ps, pe, screen result, exact result)
#ifdef PS_STAMP
    long long t0, ph[12], tbeg;
#endif
};
// In-kernel phase stamps (diagnostic build only, -DPS_STAMP): thread 0 accumulates s_memtime
// deltas per phase of the window scan; never compiled into the product library.
#ifdef PS_STAMP
#define PS_STAMP_AT(wk, i) do { if (threadIdx.x == 0) { long long t_ = clock64(); (wk).ph[i] += t_ - (wk).t0; (wk).t0 = t_; } } while (0)
#define PS_WORK_INIT {0, 0, 0, 0, {0, 0, 0, 0}, clock64(), {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, clock64()}
#else
#define PS_STAMP_AT(wk, i) do { } while (0)
#define PS_WORK_INIT {0, 0, 0, 0, {0, 0, 0, 0}}
#endif

// ---- sample access --------------------------------------------------------------------------
// DT of the kernel templates: bit 0 = sample type (PS_DTYPE_F32 / PS_DTYPE_I16), bit 2 = wide K0 digest (seg_bs.hpp:
// 64-bit integer block sums for counts up to 2^23 from the event's first sample; block-sum scan kernels only)
constexpr int DT_WIDE = 4;                           // (a flag bit outside the public dtype values: PS_DTYPE_F64 == 2 is a sample type of the filter kernels)
static_assert((PS_DTYPE_F32 & DT_WIDE) == 0 && (PS_DTYPE_I16 & DT_WIDE) == 0 && (PS_DTYPE_F64 & DT_WIDE) == 0, "DT_WIDE must not collide with a sample type");
constexpr int sdt(int DT) { return DT & 1; }
template <int DT> struct Raw { typedef float type; };
template <> struct Raw<PS_DTYPE_I16> { typedef int16_t type; };
template <> struct Raw<PS_DTYPE_I16 | DT_WIDE> { typedef int16_t type; };

template <int DT>
__device__ __forceinline__ int to_count(const DevCfg &c, typename Raw<DT>::type v, unsigned &bad)
{
    if (sdt(DT) == PS_DTYPE_F32) {
        const float k = static_cast<float>(v) * c.inv_q;
        const int ki = __float2int_rn(k);
        if (static_cast<float>(ki) != k || !(fabsf(k) < 8388608.f)) bad |= ST_OFF_GRID;
        return ki;
    }
    return static_cast<int>(v) + c.off_counts;
}

template <int DT>
__device__ __forceinline__ int load_count(const DevCfg &c, int64_t gi, unsigned &bad)
{
    if (sdt(DT) == PS_DTYPE_F32) {
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

// ---- wave64 primitives on DPP (no LDS traffic) -------------------------------------------------
// Hillis-Steele inclusive scan inside each row of 16 lanes (row_shr 1,2,4,8), then row_bcast15 /
// row_bcast31 carry the row totals across (gfx9 DPP controls).  Every step combines two DISJOINT
// lane ranges, so a max-with-payload reduction never sees the same element twice.  Lane 63 ends
// up holding the reduction over the whole wave.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov(int identity, int x)
{
    return __builtin_amdgcn_update_dpp(identity, x, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_movf(float identity, float x)
{
    return __int_as_float(dpp_mov<CTRL, ROW_MASK>(__float_as_int(identity), __float_as_int(x)));
}
// identity 0: with every row enabled the lanes without a source read 0 through bound_ctrl, and the compiler does not have to
// initialise the destination first (two moves less per fp64 step of the scans)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov0(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_movd(double x)          // identity 0.0
{
    const int lo = dpp_mov0<CTRL, ROW_MASK>(__double2loint(x));
    const int hi = dpp_mov0<CTRL, ROW_MASK>(__double2hiint(x));
    return __hiloint2double(hi, lo);
}
#define PS_DPP_STEPS(X) X(0x111, 0xf) X(0x112, 0xf) X(0x114, 0xf) X(0x118, 0xf) X(0x142, 0xa) X(0x143, 0xc)

__device__ __forceinline__ void wave_incl_scan2(double &a, double &b)
{
#define PS_STEP(CTRL, RM) { const double ta = dpp_movd<CTRL, RM>(a), tb = dpp_movd<CTRL, RM>(b); a += ta; b += tb; }
    PS_DPP_STEPS(PS_STEP)
#undef PS_STEP
}
__device__ __forceinline__ void wave_minmax(int &mn, int &mx)      // result in lane 63
{
#define PS_STEP(CTRL, RM) { const int tn = dpp_mov<CTRL, RM>(0x7fffffff, mn), tx = dpp_mov<CTRL, RM>(static_cast<int>(0x80000000), mx); mn = min(mn, tn); mx = max(mx, tx); }
    PS_DPP_STEPS(PS_STEP)
#undef PS_STEP
}

// ---- workgroup primitives -------------------------------------------------------------------
// The NT = 64 instantiations of everything below serve ONE WAVE, which may be one of several in its workgroup (bridge
// look-ahead): their thread index is the lane and their "barrier" is wave-local -- lanes of a wave run in lockstep, so
// LDS traffic between them only has to be complete (s_waitcnt), not synchronised with other waves.
template <int NT> __device__ __forceinline__ int ps_tid()
{
    return NT == 64 ? static_cast<int>(threadIdx.x & 63u) : static_cast<int>(threadIdx.x);
}
template <int NT> __device__ __forceinline__ void ps_sync()
{
    if constexpr (NT == 64) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// Exclusive prefix over the workgroup of two fp64 values that are exact integers (so the
// summation order does not matter); also returns the workgroup totals.  One barrier.
template <int NT>
__device__ __forceinline__ void block_exscan2(double v1, double v2, double &e1, double &e2,
                                              double &t1, double &t2, SharedT<NT> &sh)
{
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = ps_tid<NT>() >> 6;
    double i1 = v1, i2 = v2;
    wave_incl_scan2(i1, i2);
    if (lane == 63) { sh.wsum1[wave] = i1; sh.wsum2[wave] = i2; }
    ps_sync<NT>();
    double b1 = 0, b2 = 0, s1 = 0, s2 = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const double x1 = sh.wsum1[w], x2 = sh.wsum2[w];
        if (w < wave) { b1 += x1; b2 += x2; }
        s1 += x1; s2 += x2;
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

// Workgroup arg-max with the reference tie-break; result uniform.  One barrier (the slots it
// writes were last read before an earlier barrier of the same scan).
template <int NT>
__device__ __forceinline__ int block_argmax(double g, int idx, SharedT<NT> &sh, double *gain_out)
{
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = ps_tid<NT>() >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        double og = __shfl_down(g, d);
        int oi = __shfl_down(idx, d);
        if (beats(og, oi, g, idx)) { g = og; idx = oi; }
    }
    if (lane == 0) { sh.wbest[wave] = g; sh.widx[wave] = idx; }
    ps_sync<NT>();
    double bg = sh.wbest[0];
    int bi = sh.widx[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
        const double og = sh.wbest[w];
        const int oi = sh.widx[w];
        if (beats(og, oi, bg, bi)) { bg = og; bi = oi; }
    }
    if (gain_out) *gain_out = bg;
    return bi;
}

// ---- exact scan of a staged window: cparsers.pyx:157-178 to the letter -------------------------
// ys[0..n) hold the window's counts (LDS) or, when ys == nullptr, samples are read from HBM.
template <int NT, int DT>
__device__ int scan_exact(const DevCfg &c, const lds_t *ys, int64_t g0, int ps, int n, int cand_lo,
                          int cand_hi, double thresh, double *scores, SharedT<NT> &sh, unsigned &bad,
                          double *best_gain_out)
{
    const int ch = ((n + NT - 1) / NT) | 1;
    const long long lo_ll = static_cast<long long>(ps_tid<NT>()) * ch;
    const int lo = lo_ll < n ? static_cast<int>(lo_ll) : n;
    const int hi = (lo + ch < n) ? lo + ch : n;
    ps_sync<NT>();          // a screen that bailed out early may still be reading the scan slots

    double s1 = 0, s2 = 0;
    for (int j = lo; j < hi; ++j) {
        double k = static_cast<double>(ys ? ys[j] : load_count<DT>(c, g0 + j, bad));
        s1 += k; s2 += k * k;
    }
    double a1, a2, t1, t2;
    block_exscan2<NT>(s1, s2, a1, a2, t1, t2, sh);

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
        double k = static_cast<double>(ys ? ys[j] : load_count<DT>(c, g0 + j, bad));
        a1 += k; a2 += k * k;
    }
    return block_argmax<NT>(best, bi, sh, best_gain_out);
}

// ---- exact scan from the reference's OWN prefix sums: cparsers.pyx:31-38 and :157-178 to the letter -----------------
// For float64 input on no ADC grid the reference's decisions depend on the rounding of its sequential cumsums
// (c = np.cumsum(current), c2 = np.cumsum(current * current), cparsers.pyx:110-111): exact sums of a re-quantised copy decide
// near ties differently.  This route reads c and c2 as cumsum_ref_kernel left them -- the same additions in
// the same order -- and evaluates var_c with the reference's expressions: (c2[e-1] - c2[s-1]) / (e - s) - ((c[e-1] - c[s-1])
// / (e - s)) ** 2, the start == 0 branch without a subtraction, gain = var_summed - (low + high), strict '>', first maximum.
// C, C2: the event's arrays (C[i] = sum of x[0..i]).  No FMA contraction (the reference build has none); x ** 2 is pow(x, 2.0),
// which glibc rounds correctly: the same double as x * x.  The only difference left is the logarithm's last bit (section 2).
__device__ __forceinline__ double ref_var_c(const double *C, const double *C2, int start, int end)
{
#pragma clang fp contract(off)
    if (start == end) return 0.0;
    if (start == 0) {
        const double en = static_cast<double>(end);
        const double m = C[end - 1] / en;
        const double v = C2[end - 1] / en;
        const double mm = m * m;
        return v - mm;
    }
    const double dn = static_cast<double>(end - start);
    const double d2 = C2[end - 1] - C2[start - 1];
    const double d1 = C[end - 1] - C[start - 1];
    const double v = d2 / dn;
    const double m = d1 / dn;
    const double mm = m * m;
    return v - mm;
}

template <int NT>
__device__ int scan_exact_prefix(const DevCfg &c, int64_t base, int ps, int n, int cand_lo, int cand_hi, double thresh,
                                 SharedT<NT> &sh, double *best_gain_out)
{
#pragma clang fp contract(off)
    const double *C = c.pre_c + base, *C2 = c.pre_c2 + base;
    const int start = ps, end = ps + n;
    ps_sync<NT>();
    const double var_summed = static_cast<double>(n) * log(ref_var_c(C, C2, start, end));
    double best = thresh;
    int bi = -1;
    for (int i = cand_lo + static_cast<int>(ps_tid<NT>()); i <= cand_hi; i += NT) {
        const double low = static_cast<double>(i - start) * log(ref_var_c(C, C2, start, i));
        const double high = static_cast<double>(end - i) * log(ref_var_c(C, C2, i, end));
        const double sm = low + high;
        const double gain = var_summed - sm;
        if (gain > best) { best = gain; bi = i; }      // (ascending i per thread, strict '>': the thread's first maximum)
    }
    return block_argmax<NT>(best, bi, sh, best_gain_out);
}

// c = cumsum(x), c2 = cumsum(x * x) of every event, exactly as numpy forms them: one strictly sequential chain of fp64
// additions per event (add.accumulate; np.multiply rounds the squares first).  One wave per event: the lanes move 512 values
// at a time through LDS, lane 0 runs the two chains.  ~10-20 ns per sample and event, events in parallel: the price of the
// reference's own rounding -- this route is for events the fast route flags (near ties) or for callers who ask for it.
__global__ __launch_bounds__(64) void cumsum_ref_kernel(const double *__restrict__ x, const int64_t *__restrict__ ev_start,
                                                        const int64_t *__restrict__ ev_len, int n_ev, double *C, double *C2)
{
#pragma clang fp contract(off)
    constexpr int TRIP = 512;
    __shared__ double xs[TRIP], cs[TRIP], c2s[TRIP];
    const int lane = threadIdx.x;
    for (int e = blockIdx.x; e < n_ev; e += gridDim.x) {
        const int64_t base = ev_start[e], len = ev_len[e];
        double c = 0.0, c2 = 0.0;
        for (int64_t b = 0; b < len; b += TRIP) {
            const int m = static_cast<int>(len - b < TRIP ? len - b : TRIP);
            for (int k = lane; k < m; k += 64) xs[k] = x[base + b + k];
            ps_sync<64>();
            if (lane == 0) {
#pragma unroll 8
                for (int j = 0; j < m; ++j) {
                    const double v = xs[j];
                    const double vv = v * v;
                    c = c + v;                                // (numpy's first element is x[0] itself: 0.0 + x[0], the same double -- but for the
                    c2 = c2 + vv;                             //  sign of a zero, which no later expression sees)
                    cs[j] = c; c2s[j] = c2;
                }
            }
            ps_sync<64>();
            for (int k = lane; k < m; k += 64) { C[base + b + k] = cs[k]; C2[base + b + k] = c2s[k]; }
            ps_sync<64>();
        }
    }
}

// ---- fp32 screen ---------------------------------------------------------------------------------
// Bound on |screened gain - reference gain| in log2 units for a window of n samples
// (DESIGN.md "error budget of the screen"): per side, D = nL*p2 - p1^2 carries <= 3*eps*kappa
// (kappa <= 4 enforced), two rcp and two products 6*eps, v_log_f32 <= 1 ulp at |log2| < 32,
// the subtraction of c0 and the two final products ~3*eps*|term|; all times nL+nR = n.
__device__ __forceinline__ float screen_delta_log2(int n) { return 0.02f + 8.0e-6f * static_cast<float>(n); }

struct Top2 { float b, s; int i; };

typedef float f2 __attribute__((ext_vector_type(2)));

// Screened gain (log2 units, relative to the whole window) of one candidate from the exact
// integer moment sums about the chunk's centres: left (p1 = sum z, p2 = sum z^2, nl samples),
// right (r1, r2, nr).  The (left, right) pair is carried in 2-wide vectors so the arithmetic
// maps to v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32.  `guard` accumulates the validity margins
// (kappa = n*p2/D <= 4 and the variance floor) with v_min3_f32.
__device__ __forceinline__ float screen_gain(int p1, unsigned p2, int r1, unsigned r2, f2 nv, f2 cc,
                                             float &kguard, float &umin)
{
    const f2 s1 = {static_cast<float>(p1), static_cast<float>(r1)};
    const f2 s2 = {static_cast<float>(p2), static_cast<float>(r2)};
    const f2 ns2 = nv * s2;
    const f2 D = __builtin_elementwise_fma(-s1, s1, ns2);                 // n*p2 - p1^2
    const f2 four = {4.0f, 4.0f};
    const f2 g = __builtin_elementwise_fma(four, D, -ns2);                // >= 0  <=>  kappa <= 4
    const f2 r = {__builtin_amdgcn_rcpf(nv.x), __builtin_amdgcn_rcpf(nv.y)};
    const f2 u = D * r * r;                                               // variances (counts^2)
    kguard = fminf(kguard, fminf(g.x, g.y));
    umin = fminf(umin, fminf(u.x, u.y));
    const f2 lg = {__builtin_amdgcn_logf(u.x), __builtin_amdgcn_logf(u.y)};
    const f2 t = nv * (lg - cc);
    return -(t.x + t.y);
}

__device__ __forceinline__ void top2_push(Top2 &t, float g, int i)
{
    // second largest of {b, s, g} with s <= b is their median; an exact tie gives s == b
    t.s = __builtin_amdgcn_fmed3f(t.b, g, t.s);
    t.i = g > t.b ? i : t.i;
    t.b = fmaxf(t.b, g);
}
__device__ __forceinline__ void top2_merge(Top2 &t, float ob, float os, int oi)
{
    // equal values at different indices must register as a tie: second == best
    if (ob > t.b) { t.s = fmaxf(t.b, os); t.b = ob; t.i = oi; }
    else t.s = fmaxf(t.s, ob);
}

// Returns 1 if the screen produced a final answer (result in *split), 0 if the exact path is
// needed.  All threads return the same value.
template <int NT>
__device__ int scan_screen(const DevCfg &c, const lds_t *ys, int ps, int n, int cand_lo, int cand_hi,
                           double thresh, int kmin, int kmax, SharedT<NT> &sh, int *split, Work &wk)
{
    constexpr int NW = NT / 64;
    const int ch = ((n + NT - 1) / NT) | 1;           // odd lane stride: conflict-free ds_read_b32
    const int lo = min(n, static_cast<int>(threadIdx.x) * ch);
    const int hi = min(n, lo + ch);
    const int R = kmax - kmin;                        // |k - centre| <= R for any centre in [kmin,kmax]
    // integer moment sums must stay exact: chunk sums below 2^31, squares below 2^30
    if (R >= 32768 || static_cast<long long>(ch + 1) * (static_cast<long long>(R) * R) >= (1LL << 31)) return 0;
    const int m0 = kmin + (R >> 1);

    int s1 = 0;
    unsigned s2 = 0;
    for (int rep = 0; rep < c.rep_sum; ++rep) { s1 = 0; s2 = 0;
    for (int j = lo; j < hi; ++j) {
        const int y = ys[j] - m0;
        s1 += y;
        s2 += static_cast<unsigned>(__mul24(y, y));
    } asm volatile("" :: "v"(s1), "v"(s2)); }
    PS_STAMP_AT(wk, 2);                                // chunk sums from LDS
    double a1, a2, t1, t2;
    block_exscan2<NT>(static_cast<double>(s1), static_cast<double>(s2), a1, a2, t1, t2, sh);
    PS_STAMP_AT(wk, 3);                                // workgroup scan

    const double dn = static_cast<double>(n);
    const double Dtot = dn * t2 - t1 * t1;            // n^2 * variance (shift invariant)
    if (!(Dtot > 0.0)) return 0;                      // zero-variance window: reference inf/NaN path
    // log2 of the window variance (counts^2), identical in all threads: 4 roundings + 1 ulp of log
    const float rn = __builtin_amdgcn_rcpf(static_cast<float>(n));
    const float c0 = __builtin_amdgcn_logf(static_cast<float>(Dtot) * rn * rn);
    const f2 cc = {c0, c0};
    // candidates of this thread: window-local j in [clo, chi)
    const int clo = max(lo, cand_lo - ps), chi = min(hi, cand_hi - ps + 1);
    Top2 top = {-INFINITY, -INFINITY, -1};
    unsigned flag = 0;
    float kguard = INFINITY, umin = INFINITY;         // validity margins, minimum over candidates
    if (clo < chi) {
        for (int j = lo; j < clo; ++j) {              // advance the exact prefix to the first candidate
            const double y = static_cast<double>(ys[j] - m0);
            a1 += y; a2 += y * y;
        }
        // Per-chunk re-centring on (integers near) the left and right means: exact integer sums of
        // z = k - c and z^2 per side -> D = n*p2 - p1^2 has no cancellation.  The first moments fit
        // int32 (|a1| <= n*R/2 < 2^31), the second moments are formed in fp64 (exact below 2^53).
        const int nl0 = clo, nr0 = n - clo;
        const double b1 = t1 - a1, b2 = t2 - a2;
        const int a1i = static_cast<int>(a1), b1i = static_cast<int>(b1);
        const int muL = __float2int_rn(static_cast<float>(a1i) * __builtin_amdgcn_rcpf(static_cast<float>(nl0)));
        const int muR = __float2int_rn(static_cast<float>(b1i) * __builtin_amdgcn_rcpf(static_cast<float>(nr0)));
        int p1 = a1i - __mul24(nl0, muL), r1 = b1i - __mul24(nr0, muR);
        const double muLd = static_cast<double>(muL), muRd = static_cast<double>(muR);
        const double p2d = a2 - muLd * (a1 + static_cast<double>(p1));     // a2 - mu*(2*a1 - nl0*mu)
        const double r2d = b2 - muRd * (b1 + static_cast<double>(r1));
        const float roomf = static_cast<float>(chi - clo) * (static_cast<float>(R) + 1.0f) * (static_cast<float>(R) + 1.0f);
        if (!(static_cast<float>(p2d) + roomf < 4.2e9f) || !(static_cast<float>(r2d) < 4.2e9f) ||
            !(fabsf(static_cast<float>(p1)) + roomf < 2.1e9f) || !(fabsf(static_cast<float>(r1)) + roomf < 2.1e9f))
            flag = 1;
        unsigned p2 = static_cast<unsigned>(p2d), r2 = static_cast<unsigned>(r2d);
        const int cL = m0 + muL, cR = m0 + muR;
        f2 nv = {static_cast<float>(clo), static_cast<float>(n - clo)};
        const f2 step1 = {1.0f, -1.0f};
        // reference noise floor: below this variance (counts^2) the reference's own
        // c2/n - (c/n)^2 loses more than ~1e-3 of gain to cancellation -> decide exactly
        const float mabs = fmaxf(fabsf(static_cast<float>(kmin)), fabsf(static_cast<float>(kmax)));
        const float vfloor = mabs * mabs * 1.0e-9f;
        const int p1_0 = p1, r1_0 = r1; const unsigned p2_0 = p2, r2_0 = r2; const f2 nv_0 = nv;
        if (!flag) for (int rep = 0; rep < c.rep_eval; ++rep) {
            p1 = p1_0; r1 = r1_0; p2 = p2_0; r2 = r2_0; nv = nv_0; top.b = -INFINITY; top.s = -INFINITY; top.i = -1;
            asm volatile("" : "+v"(p1), "+v"(r1));
            // 4 candidates per trip: the integer moment recurrences are sequential but cheap,
            // the four float pipelines (rcp, log2, fma) are independent and overlap.
            int j = clo;
            for (; j + 4 <= chi; j += 4) {
                int q1[4], q3[4];
                unsigned q2[4], q4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    q1[u] = p1; q2[u] = p2; q3[u] = r1; q4[u] = r2;
                    const int k = ys[j + u];
                    const int zl = k - cL, zr = k - cR;
                    p1 += zl; p2 += static_cast<unsigned>(__mul24(zl, zl));
                    r1 -= zr; r2 -= static_cast<unsigned>(__mul24(zr, zr));
                }
                float g[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    g[u] = screen_gain(q1[u], q2[u], q3[u], q4[u], nv, cc, kguard, umin);
                    nv += step1;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) top2_push(top, g[u], j + u);
            }
            for (; j < chi; ++j) {
                const float g = screen_gain(p1, p2, r1, r2, nv, cc, kguard, umin);
                top2_push(top, g, j);
                const int k = ys[j];
                const int zl = k - cL, zr = k - cR;
                p1 += zl; p2 += static_cast<unsigned>(__mul24(zl, zl));
                r1 -= zr; r2 -= static_cast<unsigned>(__mul24(zr, zr));
                nv += step1;
            }
            if (!(kguard >= 0.0f) || !(umin >= vfloor)) flag = 1;   // also catches NaN
        }
    }
    PS_STAMP_AT(wk, 4);                                // per-thread setup + candidate loop
    // workgroup top-2 + flags
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#define PS_STEP(CTRL, RM) { const float ob = dpp_movf<CTRL, RM>(-INFINITY, top.b), os = dpp_movf<CTRL, RM>(-INFINITY, top.s); \
                            const int oi = dpp_mov<CTRL, RM>(-1, top.i); const int of = dpp_mov<CTRL, RM>(0, static_cast<int>(flag)); \
                            top2_merge(top, ob, os, oi); flag |= static_cast<unsigned>(of); }
    PS_DPP_STEPS(PS_STEP)
#undef PS_STEP
    if (lane == 63) { sh.fbest[wave] = top.b; sh.fsecond[wave] = top.s; sh.fidx[wave] = top.i; sh.wflag[wave] = flag; }
    __syncthreads();
    Top2 all = {sh.fbest[0], sh.fsecond[0], sh.fidx[0]};
    unsigned anyflag = sh.wflag[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
        top2_merge(all, sh.fbest[w], sh.fsecond[w], sh.fidx[w]);
        anyflag |= sh.wflag[w];
    }
    PS_STAMP_AT(wk, 5);                                // top-2 reduce + barrier
    if (anyflag) return 0;
    const float dlt = screen_delta_log2(n);
    const float LN2 = 0.6931471805599453f;
    const float thr_log2 = static_cast<float>(thresh * 1.4426950408889634);
    const float dthr = dlt + 3.0e-6f * static_cast<float>(n) + 1.0e-6f * fabsf(thr_log2);   // + error of c0, of thr_log2
    (void)LN2;
    if (all.b < thr_log2 - dthr) { *split = -1; return 1; }                // certainly no candidate above min_gain
    if (all.b > thr_log2 + dthr && all.s < all.b - 2.0f * dlt) { *split = ps + all.i; return 1; }   // unique, clear winner
    return 0;
}


// ---- screen for windows whose counts do not fit the int16 LDS image ------------------------------------------
// A filtered event (Event.filter) is float64 off every ADC grid; Event.parse rounds it to a 2^-18 pA grid, which makes
// counts of +-2^21.  Such a window used to go straight to scan_exact: two fp64 logs and four fp64 divisions per
// candidate.  This screen reads the samples from HBM like scan_exact does (each thread owns a contiguous chunk), but
// evaluates the candidates like the block-sum scan: sums of y = k - m0 (m0 = middle of the window's range) and y^2 in
// fp64, D = n*S2 - S1^2 per side in fp64, everything after the conversion of D in fp32 (v_rcp_f32, v_log_f32).
// Accuracy: the fp64 sums carry a relative error of at most (chunk + log2 NT) * 2^-53 ~ 2^-47, D loses
// kappa = n*S2/D of that, and kappa <= (R/2)^2 / V is kept below 2^22 by the variance floor: D is good to 2^-25, the
// rest of the budget is that of the block-sum screen (delta(n) = 0.02 + 8e-6 n log2 units).  Decisions: the same
// threshold band / unique-winner rule; anything else is decided by scan_exact.
template <int NT, int DT>
__device__ int scan_screen_wide(const DevCfg &c, int64_t g0, int ps, int n, int cand_lo, int cand_hi, double thresh,
                                int kmin, int kmax, SharedT<NT> &sh, int *split, unsigned &bad)
{
    constexpr int NW = NT / 64;
    const int ch = ((n + NT - 1) / NT) | 1;
    const int lo = min(n, static_cast<int>(ps_tid<NT>()) * ch);
    const int hi = min(n, lo + ch);
    const long long R = static_cast<long long>(kmax) - kmin;
    if (R >= (1LL << 24)) return 0;                   // (n * (R/2)^2 must stay below 2^60 or so: fp64 has the room, stay modest)
    const int m0 = kmin + static_cast<int>(R >> 1);
    double s1 = 0, s2 = 0;
    for (int j = lo; j < hi; ++j) {
        const double y = static_cast<double>(load_count<DT>(c, g0 + j, bad) - m0);
        s1 += y; s2 = fma(y, y, s2);
    }
    double a1, a2, t1, t2;
    block_exscan2<NT>(s1, s2, a1, a2, t1, t2, sh);
    const double dn = static_cast<double>(n);
    const double Dtot = dn * t2 - t1 * t1;
    if (!(Dtot > 0.0)) return 0;
    const float rn = __builtin_amdgcn_rcpf(static_cast<float>(n));
    const float c0 = __builtin_amdgcn_logf(static_cast<float>(Dtot) * rn * rn);
    const f2 cc = {c0, c0};
    const float mabs = fmaxf(fabsf(static_cast<float>(kmin)), fabsf(static_cast<float>(kmax)));
    const float half = 0.5f * static_cast<float>(R);
    const float vfloor = fmaxf(mabs * mabs * 1.0e-9f, half * half * 2.4e-7f);       // 2^-22
    const int clo = max(lo, cand_lo - ps), chi = min(hi, cand_hi - ps + 1);
    Top2 top = {-INFINITY, -INFINITY, -1};
    unsigned flag = 0;
    if (clo < chi) {
        for (int j = lo; j < clo; ++j) {
            const double y = static_cast<double>(load_count<DT>(c, g0 + j, bad) - m0);
            a1 += y; a2 = fma(y, y, a2);
        }
        float umin = INFINITY;
        for (int j = clo; j < chi; ++j) {
            const double nl = static_cast<double>(j), nr = dn - nl;
            const double b1 = t1 - a1, b2 = t2 - a2;
            const double DL = fma(nl, a2, -(a1 * a1)), DR = fma(nr, b2, -(b1 * b1));
            const f2 D = {static_cast<float>(DL), static_cast<float>(DR)};
            const f2 nv = {static_cast<float>(j), static_cast<float>(n - j)};
            const f2 r = {__builtin_amdgcn_rcpf(nv.x), __builtin_amdgcn_rcpf(nv.y)};
            const f2 u = D * r * r;
            umin = fminf(umin, fminf(u.x, u.y));
            const f2 lg = {__builtin_amdgcn_logf(u.x), __builtin_amdgcn_logf(u.y)};
            const f2 t = nv * (lg - cc);
            top2_push(top, -(t.x + t.y), j);
            const double y = static_cast<double>(load_count<DT>(c, g0 + j, bad) - m0);
            a1 += y; a2 = fma(y, y, a2);
        }
        if (!(umin >= vfloor)) flag = 1;              // also catches NaN
    }
    const int lane = threadIdx.x & 63, wave = ps_tid<NT>() >> 6;
#define PS_STEP(CTRL, RM) { const float ob = dpp_movf<CTRL, RM>(-INFINITY, top.b), os = dpp_movf<CTRL, RM>(-INFINITY, top.s); \
                            const int oi = dpp_mov<CTRL, RM>(-1, top.i); const int of = dpp_mov<CTRL, RM>(0, static_cast<int>(flag)); \
                            top2_merge(top, ob, os, oi); flag |= static_cast<unsigned>(of); }
    PS_DPP_STEPS(PS_STEP)
#undef PS_STEP
    ps_sync<NT>();                                     // (block_exscan2's slots are not reused here, but fbest.. may be by a caller)
    if (lane == 63) { sh.fbest[wave] = top.b; sh.fsecond[wave] = top.s; sh.fidx[wave] = top.i; sh.wflag[wave] = flag; }
    ps_sync<NT>();
    Top2 all = {sh.fbest[0], sh.fsecond[0], sh.fidx[0]};
    unsigned anyflag = sh.wflag[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
        top2_merge(all, sh.fbest[w], sh.fsecond[w], sh.fidx[w]);
        anyflag |= sh.wflag[w];
    }
    ps_sync<NT>();
    if (anyflag) return 0;
    const float dlt = screen_delta_log2(n);
    const float thr_log2 = static_cast<float>(thresh * 1.4426950408889634);
    const float dthr = dlt + 3.0e-6f * static_cast<float>(n) + 1.0e-6f * fabsf(thr_log2);
    if (all.b < thr_log2 - dthr) { *split = -1; return 1; }
    if (all.b > thr_log2 + dthr && all.s < all.b - 2.0f * dlt) { *split = ps + all.i; return 1; }
    return 0;
}


// ---- fp32 screen with exact block pruning -----------------------------------------------------------
// Most candidates never need an evaluation.  For a block of candidates j in [jb, je) the sums of
// squared deviations are monotone: SS_L(j) >= SS_L(jb) (the left part only grows) and
// SS_R(j) >= SS_R(je) (the right part only shrinks), hence with a = log2 V_L(jb) - c0,
// b = log2 V_R(je) - c0 and log2(1+x) <= x*log2(e):
//     G(j) <= -( nl*(a - (nl-nl_b)*log2e/nl_b) + nr*(b - (nr-nr_e)*log2e/nr_e) ),   nl = j, nr = n-j,
// a convex function of nl, so its maximum over the block is at one of the two end candidates.
// Only the block-boundary candidates are evaluated (one in PBLK); a block whose bound cannot reach
// T0 = max(threshold, best boundary gain seen by the first round) minus the error margins is
// discarded, the few others are queued in LDS and evaluated candidate by candidate by the whole
// workgroup.  Decisions are the same as with every candidate evaluated: a discarded candidate is
// provably below the threshold band or more than 2*delta below the winner.
struct ScrOut { float g; f2 lg; f2 r; };

__device__ __forceinline__ ScrOut screen_eval(int p1, unsigned p2, int r1, unsigned r2, f2 nv, f2 cc,
                                              float &kguard, float &umin)
{
    const f2 s1 = {static_cast<float>(p1), static_cast<float>(r1)};
    const f2 s2 = {static_cast<float>(p2), static_cast<float>(r2)};
    const f2 ns2 = nv * s2;
    const f2 D = __builtin_elementwise_fma(-s1, s1, ns2);
    const f2 four = {4.0f, 4.0f};
    const f2 g = __builtin_elementwise_fma(four, D, -ns2);
    const f2 r = {__builtin_amdgcn_rcpf(nv.x), __builtin_amdgcn_rcpf(nv.y)};
    const f2 u = D * r * r;
    kguard = fminf(kguard, fminf(g.x, g.y));
    umin = fminf(umin, fminf(u.x, u.y));
    const f2 lgu = {__builtin_amdgcn_logf(u.x), __builtin_amdgcn_logf(u.y)};
    ScrOut o;
    o.lg = lgu - cc;
    o.r = r;
    const f2 t = nv * o.lg;
    o.g = -(t.x + t.y);
    return o;
}

template <int NT>
__device__ int scan_screen_pruned(const DevCfg &c, const lds_t *ys, int2 *bsum, int ps, int n, int cand_lo, int cand_hi,
                                  double thresh, int kmin, int kmax, SharedT<NT> &sh, int *split, Work &wk)
{
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ch = (((n + NT - 1) / NT) + PBLK - 1) / PBLK * PBLK + 1;   // odd stride: conflict-free reads
    const int lo = min(n, static_cast<int>(threadIdx.x) * ch);
    const int hi = min(n, lo + ch);
    const int R = kmax - kmin;
    if (R >= 23000 || static_cast<long long>(ch + 1) * (static_cast<long long>(R) * R) >= (1LL << 31)) return 0;
    const int m0 = kmin + (R >> 1);

    // pass A: per-block (PBLK samples) sums of y = k - m0 and y^2, kept in LDS for the block advance
    // of pass 1, and the chunk totals for the workgroup scan
    int s1 = 0;
    unsigned s2 = 0;
    const int kb = (ch - 1) / PBLK;                    // block slots per thread
    int2 *mybs = bsum + static_cast<int>(threadIdx.x) * kb;
    {
        int j = lo, b = 0;
        for (; j + PBLK <= hi; j += PBLK, ++b) {
            int y[PBLK];
#pragma unroll
            for (int u = 0; u < PBLK; ++u) y[u] = ys[j + u] - m0;
            int t1b = 0;
            unsigned t2b = 0;
#pragma unroll
            for (int u = 0; u < PBLK; ++u) { t1b += y[u]; t2b += static_cast<unsigned>(__mul24(y[u], y[u])); }
            mybs[b] = make_int2(t1b, static_cast<int>(t2b));
            s1 += t1b; s2 += t2b;
        }
        for (; j < hi; ++j) {
            const int y = ys[j] - m0;
            s1 += y;
            s2 += static_cast<unsigned>(__mul24(y, y));
        }
    }
    double a1, a2, t1, t2;
    block_exscan2<NT>(static_cast<double>(s1), static_cast<double>(s2), a1, a2, t1, t2, sh);

    const double dn = static_cast<double>(n);
    const double Dtot = dn * t2 - t1 * t1;
    if (!(Dtot > 0.0)) return 0;
    const float rn = __builtin_amdgcn_rcpf(static_cast<float>(n));
    const float c0 = __builtin_amdgcn_logf(static_cast<float>(Dtot) * rn * rn);
    const f2 cc = {c0, c0};
    const float dlt = screen_delta_log2(n);
    const float thr_log2 = static_cast<float>(thresh * 1.4426950408889634);
    const float dthr = dlt + 3.0e-6f * static_cast<float>(n) + 1.0e-6f * fabsf(thr_log2);
    const float nf = static_cast<float>(n);
    const float LOG2E = 1.4426950408889634f;

    const int clo = max(lo, cand_lo - ps), chi = min(hi, cand_hi - ps + 1);
    const bool have = clo < chi;
    Top2 top = {-INFINITY, -INFINITY, -1};
    unsigned flag = 0;
    float kguard = INFINITY, umin = INFINITY;
    int p1 = 0, r1 = 0, cL = 0, cR = 0;
    unsigned p2 = 0, r2 = 0;
    const float mabs = fmaxf(fabsf(static_cast<float>(kmin)), fabsf(static_cast<float>(kmax)));
    const float vfloor = mabs * mabs * 1.0e-9f;
    ScrOut b0 = {-INFINITY, {0.f, 0.f}, {0.f, 0.f}};
    if (have) {
        for (int j = lo; j < clo; ++j) {
            const double y = static_cast<double>(ys[j] - m0);
            a1 += y; a2 += y * y;
        }
        const int nl0 = clo, nr0 = n - clo;
        const double b1 = t1 - a1, b2 = t2 - a2;
        const int a1i = static_cast<int>(a1), b1i = static_cast<int>(b1);
        const int muL = __float2int_rn(static_cast<float>(a1i) * __builtin_amdgcn_rcpf(static_cast<float>(nl0)));
        const int muR = __float2int_rn(static_cast<float>(b1i) * __builtin_amdgcn_rcpf(static_cast<float>(nr0)));
        p1 = a1i - __mul24(nl0, muL); r1 = b1i - __mul24(nr0, muR);
        const double p2d = a2 - static_cast<double>(muL) * (a1 + static_cast<double>(p1));
        const double r2d = b2 - static_cast<double>(muR) * (b1 + static_cast<double>(r1));
        const float roomf = static_cast<float>(chi - clo) * (static_cast<float>(R) + 1.0f) * (static_cast<float>(R) + 1.0f);
        if (!(static_cast<float>(p2d) + roomf < 4.2e9f) || !(static_cast<float>(r2d) < 4.2e9f) ||
            !(fabsf(static_cast<float>(p1)) + roomf < 2.1e9f) || !(fabsf(static_cast<float>(r1)) + roomf < 2.1e9f))
            flag = 1;
        p2 = static_cast<unsigned>(p2d); r2 = static_cast<unsigned>(r2d);
        cL = m0 + muL; cR = m0 + muR;
        if (!flag) {
            const f2 nv = {static_cast<float>(clo), static_cast<float>(n - clo)};
            b0 = screen_eval(p1, p2, r1, r2, nv, cc, kguard, umin);
            top2_push(top, b0.g, clo);
        }
    }
    // first round: the best gain among one boundary candidate per thread fixes the pruning level
    float bm = b0.g;
#define PS_STEP(CTRL, RM) { bm = fmaxf(bm, dpp_movf<CTRL, RM>(-INFINITY, bm)); }
    PS_DPP_STEPS(PS_STEP)
#undef PS_STEP
    if (lane == 63) sh.wmaxf[wave] = bm;
    if (threadIdx.x == 0) sh.qn = 0;
    __syncthreads();
    bm = sh.wmaxf[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) bm = fmaxf(bm, sh.wmaxf[w]);
    // discard a block when its bound (+ its own rounding error) is below T0
    const float T0 = fmaxf(thr_log2 - dthr, bm - 2.0f * dlt) - 2.0f * dlt;

    if (have && !flag) {
        int j = clo;
        float aL = b0.lg.x, rlb = b0.r.x;                 // left quantities at the block start
        int bp1 = p1, br1 = r1;                           // moment sums at the block start
        unsigned bp2 = p2, br2 = r2;
        while (j < chi) {
            const int je = min(j + PBLK, chi);
            if (je - j == PBLK && ((j - lo) & (PBLK - 1)) == 0) {
                // full block on the pass-A grid: shift its sums from m0 to the side centres.
                // sum(y-mu) = S1 - 8mu ; sum(y-mu)^2 = S2 - mu*(2*S1 - 8mu)  (exact modulo 2^32)
                const int2 bs = mybs[(j - lo) / PBLK];
                const int muL = cL - m0, muR = cR - m0;
                p1 += bs.x - PBLK * muL;
                p2 += static_cast<unsigned>(bs.y) - static_cast<unsigned>(muL) * static_cast<unsigned>(2 * bs.x - PBLK * muL);
                r1 -= bs.x - PBLK * muR;
                r2 -= static_cast<unsigned>(bs.y) - static_cast<unsigned>(muR) * static_cast<unsigned>(2 * bs.x - PBLK * muR);
            } else {
                for (int m = j; m < je; ++m) {            // partial block: sample by sample
                    const int k = ys[m];
                    const int zl = k - cL, zr = k - cR;
                    p1 += zl; p2 += static_cast<unsigned>(__mul24(zl, zl));
                    r1 -= zr; r2 -= static_cast<unsigned>(__mul24(zr, zr));
                }
            }
            const f2 nv = {static_cast<float>(je), static_cast<float>(n - je)};
            ScrOut e;
            if (je < chi) {                               // next boundary: a candidate of this thread
                e = screen_eval(p1, p2, r1, r2, nv, cc, kguard, umin);
                top2_push(top, e.g, je);
            } else if (n - je >= 1) {                     // end of the thread's range: right side only matters
                float kg2 = INFINITY, um2 = INFINITY;
                e = screen_eval(p1, p2, r1, r2, nv, cc, kg2, um2);
                if (!(kg2 >= 0.0f) || !(um2 >= vfloor)) e.lg.y = -INFINITY;      // unusable: keep the block
            } else {
                e.lg.y = -INFINITY; e.r.y = 0.0f; e.lg.x = 0.f; e.r.x = 0.f; e.g = -INFINITY;
            }
            const int cnt = je - j;
            if (cnt > 1) {
                const float nl0f = static_cast<float>(j), nlef = static_cast<float>(je - 1);
                const float nr0f = nf - nl0f, nref = nf - nlef;
                const float bR = e.lg.y, rre = e.r.y;
                const float cR0 = bR - static_cast<float>(cnt) * LOG2E * rre;          // nr - nr_e = cnt at j
                const float cR1 = bR - LOG2E * rre;                                     // nr - nr_e = 1 at je-1
                const float cL1 = aL - static_cast<float>(cnt - 1) * LOG2E * rlb;
                const float h0 = -fmaf(nl0f, aL, nr0f * cR0);
                const float h1 = -fmaf(nlef, cL1, nref * cR1);
                const float U = fmaxf(h0, h1);
                if (!(U < T0)) {                          // cannot be discarded: queue for full evaluation
                    const int slot = atomicAdd(&sh.qn, 1);
                    if (slot < SharedT<NT>::QN) {
                        QEnt q;
                        q.j = j; q.jend = je; q.p1 = bp1; q.r1 = br1; q.p2 = bp2; q.r2 = br2; q.cL = cL; q.cR = cR;
                        sh.q[slot] = q;
                    }
                }
            }
            aL = e.lg.x; rlb = e.r.x;
            bp1 = p1; br1 = r1; bp2 = p2; br2 = r2;
            j = je;
        }
    }
    __syncthreads();
    const int qn = sh.qn;
    if (qn > SharedT<NT>::QN) flag = 1;                              // bound too weak on this window: decide exactly
    else {
        for (int idx = threadIdx.x; idx < qn * (PBLK - 1); idx += NT) {
            const int e = idx / (PBLK - 1), t = idx - e * (PBLK - 1) + 1;
            const QEnt q = sh.q[e];
            const int j = q.j + t;
            if (j < q.jend) {
                int q1 = q.p1, q3 = q.r1;
                unsigned q2 = q.p2, q4 = q.r2;
                for (int m = q.j; m < j; ++m) {
                    const int k = ys[m];
                    const int zl = k - q.cL, zr = k - q.cR;
                    q1 += zl; q2 += static_cast<unsigned>(__mul24(zl, zl));
                    q3 -= zr; q4 -= static_cast<unsigned>(__mul24(zr, zr));
                }
                const f2 nv = {static_cast<float>(j), static_cast<float>(n - j)};
                const ScrOut o = screen_eval(q1, q2, q3, q4, nv, cc, kguard, umin);
                top2_push(top, o.g, j);
            }
        }
    }
    if (!(kguard >= 0.0f) || !(umin >= vfloor)) flag = 1;
    // workgroup top-2 + flags
#define PS_STEP(CTRL, RM) { const float ob = dpp_movf<CTRL, RM>(-INFINITY, top.b), os = dpp_movf<CTRL, RM>(-INFINITY, top.s); \
                            const int oi = dpp_mov<CTRL, RM>(-1, top.i); const int of = dpp_mov<CTRL, RM>(0, static_cast<int>(flag)); \
                            top2_merge(top, ob, os, oi); flag |= static_cast<unsigned>(of); }
    PS_DPP_STEPS(PS_STEP)
#undef PS_STEP
    if (lane == 63) { sh.fbest[wave] = top.b; sh.fsecond[wave] = top.s; sh.fidx[wave] = top.i; sh.wflag[wave] = flag; }
    __syncthreads();
    Top2 all = {sh.fbest[0], sh.fsecond[0], sh.fidx[0]};
    unsigned anyflag = sh.wflag[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
        top2_merge(all, sh.fbest[w], sh.fsecond[w], sh.fidx[w]);
        anyflag |= sh.wflag[w];
    }
    if (anyflag) return 0;
    if (all.b < thr_log2 - dthr) { *split = -1; return 1; }
    if (all.b > thr_log2 + dthr && all.s < all.b - 2.0f * dlt) { *split = ps + all.i; return 1; }
    return 0;
}

}  // namespace ps

#include "seg_filter.hpp"
// Header written by assemble_tiles_kernel.  Kernels downstream of the device stitch take it as `hdr`: when
// it is non-null the item / job count is read from it on the device (no host round trip in the middle of
// the pipeline) and a failed stitch turns them into no-ops; when null the host-provided count is used.
struct AsmHeader { long long n_items, n_jobs, tscratch; int fail, pad; };
__device__ __forceinline__ long long dev_count(const AsmHeader *hdr, long long host_n)
{
    return hdr ? (hdr->fail ? 0 : hdr->n_items) : host_n;
}

#include "seg_bs.hpp"
namespace ps {

// ---- one window scan: cparsers.pyx:157-178 ------------------------------------------------------
// Window [ps, pe) of the event at `base`; candidates cand_lo..cand_hi (inclusive, event-local);
// returns the first index whose gain strictly exceeds every earlier gain and `thresh`, or -1.
template <int NT, int DT, bool VALIDATE, bool ROWSKIP = true>
__device__ int scan_window(const DevCfg &c, int *ys, int64_t base, int ps, int pe, int cand_lo, int cand_hi,
                           double thresh, double *scores, SharedT<NT> &sh, unsigned &bad, Work &wk,
                           double *best_gain_out = nullptr, int pf_end = 0, const EvRef &er = EvRef{0, 0, 0})
{
    constexpr int NW = NT / 64;
    const int n = pe - ps;
    const int64_t g0 = base + ps;
    PS_STAMP_AT(wk, 7);                                // time outside scans (recursion control, stack)
    if (ps_tid<NT>() == 0) {                           // (every wave of a multi-wave block-sum workgroup counts its own)
        wk.windows += 1;
        wk.cands += (cand_hi >= cand_lo) ? (cand_hi - cand_lo + 1) : 0;
    }
    if constexpr (NT == 64) {                          // single-wave workgroup: block-sum scan (seg_bs.hpp)
        if (c.bsum != nullptr && scores == nullptr && best_gain_out == nullptr)
            return scan_window_ph<DT, ROWSKIP>(c, er, base, ps, pe, cand_lo, cand_hi, thresh, sh, bad, wk);
    }
    if (c.pre_c != nullptr) {                          // exact route (ps_segment_exact_f64): the reference's own prefix sums, no samples read
        wk.exact += 1;
        return scan_exact_prefix<NT>(c, base, ps, n, cand_lo, cand_hi, thresh, sh, best_gain_out);
    }
    if (n > c.lds_cap) {                               // window larger than LDS: exact path from HBM
        wk.exact += 1;
        return scan_exact<NT, DT>(c, nullptr, g0, ps, n, cand_lo, cand_hi, thresh, scores, sh, bad, best_gain_out);
    }
    // 1. stage: 16-byte loads (4 fp32 / 8 int16 per lane), all of a thread's loads issued before
    //    the first use, one ds_write_b128 per 4 samples.  The staged region starts at the 16-byte
    //    boundary at or below the window start (`off` leading elements are not part of the window).
    constexpr int ES = static_cast<int>(sizeof(typename Raw<DT>::type));
    constexpr int EPV = 16 / ES;
    const char *addr0 = static_cast<const char *>(c.samples) + g0 * ES;
    const int off = static_cast<int>((reinterpret_cast<uintptr_t>(addr0) & 15u) / ES);
    const int4 *vsrc = reinterpret_cast<const int4 *>(addr0 - off * ES);
    const int nv = (n + off + EPV - 1) / EPV;          // 16-byte vectors covering the window
    lds_t *ys16 = reinterpret_cast<lds_t *>(ys);
    lds_t *ysw = ys16 + off;                           // ysw[j] = sample j of the window
    int kmin = 0x7fffffff, kmax = static_cast<int>(0x80000000);
    unsigned fracbits = 0;
    // (the previous scan's last read of ys is followed by a barrier in its reduction)
    constexpr int STAGE_U = (NT >= 1024) ? 3 : (NT >= 512 ? 5 : 10);     // covers 12k fp32 samples per trip
    for (int rep = 0; rep < c.rep_stage; ++rep)
    for (int v0 = threadIdx.x; v0 < nv; v0 += NT * STAGE_U) {
        int4 raw[STAGE_U];
        unsigned k_frac[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int u = 0; u < STAGE_U; ++u) raw[u] = vsrc[min(v0 + u * NT, nv - 1)];
        PS_STAMP_AT(wk, 8);                            // (diagnostic) loads issued
#ifdef PS_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PS_STAMP_AT(wk, 9);                            // (diagnostic) loads returned
#endif
#pragma unroll
        for (int u = 0; u < STAGE_U; ++u) {
            const int v = v0 + u * NT;
            if (v < nv) {
                int k[EPV];
                if (sdt(DT) == PS_DTYPE_F32) {
                    const float f[4] = {__int_as_float(raw[u].x), __int_as_float(raw[u].y), __int_as_float(raw[u].z),
                                        __int_as_float(raw[u].w)};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float kf = f[e] * c.inv_q;
                        k[e] = static_cast<int>(kf);          // on-grid values are integers: truncation is exact
                        // off-grid samples leave a non-zero (or NaN) remainder
                        if (VALIDATE) k_frac[e] = __float_as_uint(kf - static_cast<float>(k[e]));
                    }
                } else {
                    const int w[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        k[e] = ((e & 1) ? (w[e >> 1] >> 16) : static_cast<int>(static_cast<short>(w[e >> 1] & 0xffff))) + c.off_counts;
                }
                if (v > 0 && v < nv - 1) {                    // interior vector: every element is in the window
#pragma unroll
                    for (int e = 0; e < EPV; e += 2) {
                        kmin = min(kmin, min(k[e], k[e + 1]));     // v_min3 / v_max3
                        kmax = max(kmax, max(k[e], k[e + 1]));
                    }
                    if (VALIDATE && sdt(DT) == PS_DTYPE_F32) fracbits |= (k_frac[0] | k_frac[1]) | (k_frac[2] | k_frac[3]);
                } else {                                      // first / last vector: mask the elements outside
#pragma unroll
                    for (int e = 0; e < EPV; ++e) {
                        const bool in = v * EPV + e >= off && v * EPV + e < off + n;
                        kmin = min(kmin, in ? k[e] : 0x7fffffff);
                        kmax = max(kmax, in ? k[e] : static_cast<int>(0x80000000));
                        if (VALIDATE && sdt(DT) == PS_DTYPE_F32) fracbits |= in ? k_frac[e & 3] : 0u;
                    }
                }
#pragma unroll
                for (int e = 0; e < EPV; e += 4)        // pack to int16 (range checked below), 8 bytes per 4 samples
                    *reinterpret_cast<int2 *>(&ys16[v * EPV + e]) =
                        make_int2((k[e] & 0xffff) | (k[e + 1] << 16), (k[e + 2] & 0xffff) | (k[e + 3] << 16));
            }
        }
    }
    if (fracbits & 0x7fffffffu) bad |= ST_OFF_GRID;
    // L2 prefetch of the samples the NEXT scan of this chain will need ([pe, pf_end)): one dword
    // per 128-byte line, issued AFTER the staging data has been consumed (vmcnt returns in order,
    // so nothing of this window waits behind them) and retired only at the end of this scan:
    // the HBM latency of the next window overlaps the evaluation of this one.
    constexpr int PF_STRIDE = 128 / ES;
    int pf_val = 0;
    {
        const long long pi = static_cast<long long>(pe) + static_cast<long long>(threadIdx.x) * PF_STRIDE;
        if (pi < pf_end) pf_val = *reinterpret_cast<const int *>(static_cast<const char *>(c.samples) + ((base + pi) * ES & ~3LL));
    }
    PS_STAMP_AT(wk, 0);                                // stage: HBM loads -> LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    wave_minmax(kmin, kmax);
    if (lane == 63) { sh.wmin[wave] = kmin; sh.wmax[wave] = kmax; }
    __syncthreads();
    kmin = sh.wmin[0]; kmax = sh.wmax[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) { kmin = min(kmin, sh.wmin[w]); kmax = max(kmax, sh.wmax[w]); }
    if (sdt(DT) == PS_DTYPE_F32 && (kmin <= -8388608 || kmax >= 8388608)) bad |= ST_OFF_GRID;   // |count| >= 2^23
    if (kmin < -32768 || kmax > 32767) {               // counts do not fit the int16 LDS image: screen / exact path from HBM
        int wsplit = -1;
        const bool try_screen = c.mode != MODE_EXACT && scores == nullptr && best_gain_out == nullptr;
        const int wdone = try_screen ? scan_screen_wide<NT, DT>(c, g0, ps, n, cand_lo, cand_hi, thresh, kmin, kmax, sh, &wsplit, bad) : 0;
        if (wdone && c.mode == MODE_FAST) return wsplit;
        wk.exact += 1;
        const int ex = scan_exact<NT, DT>(c, nullptr, g0, ps, n, cand_lo, cand_hi, thresh, scores, sh, bad, best_gain_out);
        if (wdone && c.mode == MODE_VERIFY && ex != wsplit) bad |= ST_VERIFY_MISMATCH;
        return ex;
    }
    PS_STAMP_AT(wk, 1);                                // min/max reduce + barrier

    const bool want_screen = c.mode != MODE_EXACT && scores == nullptr && best_gain_out == nullptr;
    int split = -1, result;
    bool decided = false;
    if (want_screen) {
        const int done = c.prune ? scan_screen_pruned<NT>(c, ysw, reinterpret_cast<int2 *>(reinterpret_cast<char *>(ys) + (((c.lds_cap + 32) * 2 + 15) & ~15)), ps, n, cand_lo, cand_hi, thresh, kmin, kmax, sh, &split, wk)
                                 : scan_screen<NT>(c, ysw, ps, n, cand_lo, cand_hi, thresh, kmin, kmax, sh, &split, wk);
        if (done && c.mode == MODE_FAST) { result = split; decided = true; }
        else if (c.mode == MODE_VERIFY) {
            result = scan_exact<NT, DT>(c, ysw, g0, ps, n, cand_lo, cand_hi, thresh, nullptr, sh, bad, nullptr);
            if (done && result != split) bad |= ST_VERIFY_MISMATCH;
            if (!done) wk.exact += 1;
            decided = true;
        }
    }
    if (!decided) {
        wk.exact += 1;
        result = scan_exact<NT, DT>(c, ysw, g0, ps, n, cand_lo, cand_hi, thresh, scores, sh, bad, best_gain_out);
        PS_STAMP_AT(wk, 6);                            // exact rescans
    }
    asm volatile("" :: "v"(pf_val));                   // retire the prefetch load here, not earlier
    return result;
}

// ---- the window loop of _recursive_split: cparsers.pyx:186-201 ---------------------------------
// Windows j < j0 are known to hold no split (they were scanned with identical bounds by the
// parent frame, DESIGN.md "memoised left child").
template <int NT, int DT, bool VALIDATE, bool BSONLY = false, bool ROWSKIP = true>
__device__ int find_split(const DevCfg &c, int *ys, int64_t base, int start, int end, int j0, int &kind,
                          SharedT<NT> &sh, unsigned &bad, Work &wk, long long pf_lim, const EvRef &er = EvRef{0, 0, 0},
                          long long stop_lim = 0x7fffffffffffffffLL, int budget = 0x7fffffff)
{
    const long long lim = static_cast<long long>(end) - 2LL * c.mw;
    for (long long ps = static_cast<long long>(start) + static_cast<long long>(j0) * c.half; ps < lim;
         ps += c.half) {
        // A speculative tile chain gives up once its windows start beyond stop_lim: in a long stretch without
        // splits every tile would otherwise walk to the same distant anchor; the seam's bridge does that once.
        if (ps >= stop_lim || budget-- == 0) { kind = KIND_STOP; return static_cast<int>((ps - start) / c.half); }   // (windows scanned so far)
        if (ps > static_cast<long long>(start) + c.maxw) {             // :189-191
            long long a = static_cast<long long>(start) + c.maxw, b = static_cast<long long>(end) - c.mw;
            kind = KIND_EARLY;
            return static_cast<int>(a < b ? a : b);
        }
        long long pe = ps + c.W;
        if (pe > end) pe = end;                                         // :193
        int s = -1;
        if (pe - ps > 2LL * c.mw) {                                     // :164
            if constexpr (BSONLY || NT == 64) {            // (the single-wave kernels exist for the block-sum scan only)
                static_assert(NT == 64, "block-sum scan: one wave per window");
                wk.windows += 1; wk.cands += pe - ps - 2LL * c.mw + 1;       // (wave-uniform counters)
                s = scan_window_ph<DT, ROWSKIP>(c, er, base, static_cast<int>(ps), static_cast<int>(pe),
                                                static_cast<int>(ps) + c.mw, static_cast<int>(pe) - c.mw, c.min_gain, sh, bad, wk);
            } else {
                s = scan_window<NT, DT, VALIDATE, ROWSKIP>(c, ys, base, static_cast<int>(ps), static_cast<int>(pe),
                                    static_cast<int>(ps) + c.mw, static_cast<int>(pe) - c.mw,
                                    c.min_gain, nullptr, sh, bad, wk, nullptr,
                                    static_cast<int>(pe + c.W < pf_lim ? pe + c.W : pf_lim), er);
            }
        }
        if (s >= 0) { kind = KIND_HIT; return s; }                      // :195-196
    }
    if (static_cast<long long>(end) - start <= c.maxw) { kind = KIND_NONE; return -1; }   // :199-200
    long long a = static_cast<long long>(start) + c.maxw, b = static_cast<long long>(end) - c.mw;
    kind = KIND_LATE;                                                   // :201
    return static_cast<int>(a < b ? a : b);
}

// event constants of a tile job: one load per job (not per window), kept in scalar registers
__device__ __forceinline__ EvRef ev_ref_of(const DevCfg &c, int ev)
{
    EvRef r = {0, 0, 0};
    if (c.bsum != nullptr) {
        const int4 info = c.ev_info[ev];
        r.m = __builtin_amdgcn_readfirstlane(info.x);
        r.ph = __builtin_amdgcn_readfirstlane(info.y);
        r.boff = (static_cast<long long>(static_cast<unsigned>(__builtin_amdgcn_readfirstlane(info.w))) << 32) |
                 static_cast<unsigned>(__builtin_amdgcn_readfirstlane(info.z));
    }
    return r;
}

__device__ __forceinline__ int left_child_j0(int start, int split, int W, int half)
{
    long long d = static_cast<long long>(split) - W - start;
    return d < 0 ? 0 : static_cast<int>(d / half) + 1;
}

__device__ __forceinline__ void flush(unsigned bad, const Work &wk, unsigned *status, unsigned long long *work, int which = -1)
{
    if (bad) atomicOr(status, bad);
    if (threadIdx.x == 0 && wk.windows) {
        atomicAdd(&work[0], static_cast<unsigned long long>(wk.windows));
        atomicAdd(&work[1], static_cast<unsigned long long>(wk.cands));
        if (wk.exact) atomicAdd(&work[2], static_cast<unsigned long long>(wk.exact));
        if (wk.dbg[1]) { work[8] = wk.dbg[0]; work[9] = wk.dbg[1]; work[10] = wk.dbg[2]; work[11] = wk.dbg[3]; }
#ifndef PS_STAMP
        if (wk.near) atomicAdd(&work[4], static_cast<unsigned long long>(wk.near));                      // near-tie decisions
        if (which >= 0) atomicAdd(&work[17 + 3 * which], static_cast<unsigned long long>(wk.windows));   // windows per kernel
#endif
#ifdef PS_STAMP
#ifdef PS_STAMP_K
        if (which == PS_STAMP_K)
#endif
        for (int i = 0; i < 12; ++i) atomicAdd(&work[4 + i], static_cast<unsigned long long>(wk.ph[i]));
#ifdef PS_STAMP_K
        if (which == PS_STAMP_K)
#else
        if (which >= 0)
#endif
        {                                      // longest workgroup (cycles, windows) and the sum of lifetimes per kernel
            atomicMax(&work[16 + 3 * which], static_cast<unsigned long long>(clock64() - wk.tbeg));
            atomicAdd(&work[17 + 3 * which], static_cast<unsigned long long>(wk.windows));
            atomicAdd(&work[18 + 3 * which], static_cast<unsigned long long>(clock64() - wk.tbeg));
        }
#endif
    }
}

// the same for one wave of a multi-wave workgroup (every wave keeps its own counters)
__device__ __forceinline__ void flush_wave(unsigned bad, const Work &wk, unsigned *status, unsigned long long *work, int which = -1)
{
    if (bad) atomicOr(status, bad);
    if ((threadIdx.x & 63u) == 0 && wk.windows) {
        atomicAdd(&work[0], static_cast<unsigned long long>(wk.windows));
#ifndef PS_STAMP
        if (wk.near) atomicAdd(&work[4], static_cast<unsigned long long>(wk.near));
        if (which >= 0) atomicAdd(&work[17 + 3 * which], static_cast<unsigned long long>(wk.windows));
#endif
        atomicAdd(&work[1], static_cast<unsigned long long>(wk.cands));
        if (wk.exact) atomicAdd(&work[2], static_cast<unsigned long long>(wk.exact));
        if (wk.dbg[1]) { work[8] = wk.dbg[0]; work[9] = wk.dbg[1]; work[10] = wk.dbg[2]; work[11] = wk.dbg[3]; }
    }
}

// The call's small host tables (tile jobs, event offsets) come to the device by a KERNEL that reads the pinned host
// blob over PCIe and also clears the status/counter block: a hipMemcpyAsync goes through the SDMA engine, and the hand-over
// between the engines costs ~35 us before the first kernel of the call can start (memset 2 us + 18 us gap + copy 9 us +
// 8 us gap in the trace); this kernel runs back to back with its neighbours.
__global__ __launch_bounds__(256) void upload_kernel(const int4 *src, int4 *dst, long long n16, unsigned long long *zero,
                                                     int zero_words)
{
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n16; i += gridDim.x * 256LL) dst[i] = src[i];
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < zero_words; i += 256) zero[i] = 0;
}

// ---- phase 1: spine of rec(start, end), left subtrees skipped ------------------------------------
// out (private scratch, int2 = (anchor, kind)); meta[job] = (count, ended, dense position).
// After the chain stops, the workgroup reserves `count` slots in the dense list with one
// atomic and copies its anchors there so the host fetches a compact array.
template <int NT, int DT>
__global__ __launch_bounds__(NT, (NT == 64 ? PS_BS_MINW : 4)) PS_SCAN_REGS void spine_kernel(DevCfg c, const SpineJob *__restrict__ jobs, int2 *scratch,
                                                   int2 *dense, int4 *meta, unsigned long long *dense_count,
                                                   unsigned *status, unsigned long long *work, int n_jobs)
{
    extern __shared__ int ys[];
    __shared__ SharedT<NT> sh;
    if (c.bsum != nullptr && (*status & ST_WIDE_RANGE) != 0u) return;      // K0 refused the data: the host redoes the call
    unsigned bad = 0;
    Work wk = PS_WORK_INIT;
    // One workgroup per resident slot, striding over the tiles: with more tiles than slots (many short events)
    // the launch cost of a workgroup per tile would rival the scans.
    for (int jb = blockIdx.x; jb < n_jobs; jb += gridDim.x) {
        const SpineJob job = jobs[jb];
        const EvRef er = ev_ref_of(c, job.ev);
        int2 *out = scratch + job.out_off;
        int a = job.start, cnt = 0, ended = 0, flushed = 0, open_j = 0;
        for (;;) {
            int kind;
            // (device stitch only: the host stitch expects every list to reach its tile end)
            int s = find_split<NT, DT, true>(c, ys, job.base, a, job.end, 0, kind, sh, bad, wk, job.end, er,
                                             dense == nullptr ? static_cast<long long>(job.stop) + 2LL * c.W : 0x7fffffffffffffffLL);
            if (kind == KIND_NONE) { ended = 1; break; }
            if (kind == KIND_STOP) { open_j = s; break; }   // open end: the list stops short of the tile end (not "ended")
            if (cnt - flushed == SharedT<NT>::OB) {    // rare: spill the LDS buffer to the private scratch
                __syncthreads();
                for (int i = threadIdx.x; i < SharedT<NT>::OB; i += NT)
                    if (flushed + i < job.out_cap) out[flushed + i] = sh.obuf[i];
                flushed += SharedT<NT>::OB;
                __syncthreads();
            }
            if (cnt >= job.out_cap) bad |= ST_OUT_OVERFLOW;
            if (threadIdx.x == 0) sh.obuf[cnt - flushed] = make_int2(s, kind);
            ++cnt;
            a = s;
            if (a >= job.stop) break;
        }
        if (cnt > job.out_cap) cnt = job.out_cap;
        __syncthreads();
        if (dense == nullptr) {                        // device-stitch pipeline: the list stays in its own region
            for (int i = flushed + threadIdx.x; i < cnt; i += NT) out[i] = sh.obuf[i - flushed];
            if (threadIdx.x == 0) meta[jb] = make_int4(cnt, ended, open_j, 0);   // .z: windows already scanned after the last anchor
        } else {
            if (threadIdx.x == 0) {
                unsigned long long pos = atomicAdd(dense_count, static_cast<unsigned long long>(cnt));
                meta[jb] = make_int4(cnt, ended, static_cast<int>(pos), 0);
                sh.bcast = static_cast<int>(pos);
            }
            __syncthreads();
            const int pos = sh.bcast;
            for (int i = threadIdx.x; i < cnt; i += NT) dense[pos + i] = i < flushed ? out[i] : sh.obuf[i - flushed];
        }
        __syncthreads();                               // obuf is reused by the next tile
    }
    flush(bad, wk, status, work, 0);
}

// ---- phase 1b: bridge a seam -------------------------------------------------------------------
// Tile g's chain stopped at its last anchor x >= start of tile g+1.  Assuming x lies on the true
// spine (the assemble kernel checks that: tile g must be reachable), continue the chain from x
// until it produces an anchor that the downstream tile's own chain also found (or is that tile's
// start): from there on the two chains are identical.  bmeta[g] = (count, join_tile, join_idx,
// status); status 0 nothing to do, 1 joined (continue with list[join_tile][join_idx+1..]),
// 2 the chain reached the end of the event, 3 gave up (host fallback).
enum : int { BR_NONE = 0, BR_JOINED = 1, BR_ENDED = 2, BR_FAIL = 3, BR_DEFER = 4,
              BR_FAIL_REACHED = 5 };                     // (set by the stitch: a failed seam that lies on the true path)
__device__ __forceinline__ int2 bridge_at(const int2 *bridges, const int2 *ext, const int *ext_slot, int ext_stride, int g, int i)
{
    return i < BR_MAX ? bridges[static_cast<long long>(g) * BR_MAX + i]
                      : ext[static_cast<long long>(ext_slot[g]) * ext_stride + (i - BR_MAX)];
}
constexpr int BR_PATIENCE = 3;     // windows without a hit a single-wave bridge scans in one find_split before it defers
                                   // the seam to the look-ahead kernel (bmeta = (count, next window, -, BR_DEFER))

template <int NT, int DT>
__global__ __launch_bounds__(NT, (NT == 64 ? PS_BRIDGE_MINW : 4)) PS_BRIDGE_REGS void bridge_kernel(DevCfg c, const SpineJob *jobs, const int2 *lists,
                                                       const int4 *meta, int2 *bridges, int4 *bmeta,
                                                       unsigned *status, unsigned long long *work, int n_jobs, int max_single,
                                                       int budget = BR_MAX,       // anchors before the seam gives up (option bridge_budget: tests)
                                                       LatHelp lat = LAT_NONE)
{
    extern __shared__ int ys[];
    __shared__ SharedT<NT> sh;
    if (c.bsum != nullptr && (*status & ST_WIDE_RANGE) != 0u) return;      // K0 refused the data: the host redoes the call
    unsigned bad = 0;
    Work wk = PS_WORK_INIT;
    for (int g = blockIdx.x; g < n_jobs; g += gridDim.x) {
        const SpineJob job = jobs[g];
        const int4 m = meta[g];
        const bool last_tile = (g - job.first_tile) == job.ntiles - 1;
        if (last_tile || m.y != 0) {                   // chain already ran to the end of the event
            if (threadIdx.x == 0) bmeta[g] = make_int4(0, -1, 0, BR_NONE);
            continue;
        }
        // From the tile's last anchor.  A tile that gave up before its first anchor (open end) can only be entered at its
        // start: certain for an event's first tile, which is bridged from there; otherwise only if a true anchor falls
        // exactly on the tile start -- not worth a walk through the whole stretch by every such tile: it is marked failed,
        // and the stitch falls back to the host should the true chain ever reach it.
        if (m.x == 0 && g != job.first_tile) {
            if (threadIdx.x == 0) bmeta[g] = make_int4(0, -1, 0, BR_FAIL);
            continue;
        }
        int a = m.x > 0 ? lists[job.out_off + m.x - 1].x : job.start;
        int cnt = 0, st = BR_FAIL, jt = -1, ji = 0, cached = -1, ccnt = 0;
        const EvRef er = ev_ref_of(c, job.ev);
        for (int step = 0; step <= budget; ++step) {
            int u = a / job.tile_len;
            if (u > job.ntiles - 1) u = job.ntiles - 1;
            u += job.first_tile;
            const SpineJob uj = jobs[u];
            if (u != cached) {                         // cache the downstream list's positions in LDS
                ccnt = meta[u].x;
                __syncthreads();
                for (int i = threadIdx.x; i < ccnt && i < SharedT<NT>::LN; i += NT) sh.lst[i] = lists[uj.out_off + i].x;
                __syncthreads();
                cached = u;
            }
            int found = -2;                            // -1: a is the tile start, >=0: index in its list
            if (a == uj.start) found = -1;
            else {
                int lo = 0, hi = ccnt - 1;
                while (lo <= hi) {
                    const int mid = (lo + hi) >> 1;
                    const int v = mid < SharedT<NT>::LN ? sh.lst[mid] : lists[uj.out_off + mid].x;
                    if (v == a) { found = mid; break; }
                    if (v < a) lo = mid + 1; else hi = mid - 1;
                }
            }
            if (found != -2 && u != g) { st = BR_JOINED; jt = u; ji = found; break; }
            if (step == budget) break;
            // option bridge_single (default: off): hand the seam to the look-ahead kernel after max_single anchors
            // (measured slower: 0.064 -> 0.076 / 0.083 ms with 2 / 1 anchors -- most seams need no second anchor)
            if (NT == 64 && c.bsum != nullptr && cnt >= max_single) { st = BR_DEFER; jt = 0; break; }
            int kind;
            // (the samples a bridge reads were validated by the downstream tiles' own spine scans; the first call skips
            // the windows the tile's own chain already scanned without a hit before it gave up)
            const int s = find_split<NT, DT, false>(c, ys, job.base, a, job.end, step == 0 ? m.z : 0, kind, sh, bad, wk, job.end, er,
                                                    0x7fffffffffffffffLL, (NT == 64 && c.bsum != nullptr) ? BR_PATIENCE : 0x7fffffff);
            if (kind == KIND_NONE) { st = BR_ENDED; break; }
            if (kind == KIND_STOP) { st = BR_DEFER; jt = s; break; }     // a long stretch: the look-ahead kernel takes over at window s
            if (threadIdx.x == 0) sh.obuf[cnt] = make_int2(s, kind);
            ++cnt;
            a = s;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += NT) bridges[static_cast<int64_t>(g) * BR_MAX + i] = sh.obuf[i];
        if (threadIdx.x == 0) {
            bmeta[g] = make_int4(cnt, jt, ji, st);
            // a deferred seam with an open tile a chunk ahead: the stretch goes on -- the look-ahead kernel's idle workgroups stay to help
            if (st == BR_DEFER && lat.ctl != nullptr && c.half > 0) {
                const long long ps0 = static_cast<long long>(a) + static_cast<long long>(jt + LAT_W) * c.half;
                if (ps0 < static_cast<long long>(job.end) - 2LL * c.mw) {
                    int u = static_cast<int>(ps0 / job.tile_len);
                    if (u > job.ntiles - 1) u = job.ntiles - 1;
                    const int4 mu = meta[job.first_tile + u];
                    if (mu.x == 0 && mu.y == 0) atomicAdd(&lat.ctl[0], 1ull);
                }
            }
        }
        __syncthreads();                               // obuf / lst are reused by the next tile
    }
    flush(bad, wk, status, work, 1);
}

// ---- phase 1b with look-ahead (block-sum scan) ---------------------------------------------------
// A seam's chain is sequential, but between two anchors it is predictable: windows at a + j*W/2 until one hits.  LA
// waves scan LA consecutive windows of the chain at once; the first decisive outcome in window order is the chain's
// (a later window's result is simply dropped), so a stretch without splits is walked LA windows per step.  The
// bridge kernel is the place for it: few seams need more than a window or two, so the machine is nearly idle while
// the longest seam sets the kernel's duration.  Semantics: find_split (cparsers.pyx:186-201) window by window.
constexpr int BR_LA = 4;
// EXT (second chance, see EXT_MAX): the seams listed in ext_list (slot = position in the list) instead of the deferred ones;
// a seam that gave up resumes behind its last anchor, an open tile at its start (behind the windows its own chain scanned);
// anchors beyond BR_MAX go to the side buffer.
template <int DT, bool EXT = false>
__global__ __launch_bounds__(64 * BR_LA, PS_BRIDGE_MINW) PS_BRIDGE_REGS void bridge_la_kernel(DevCfg c, const SpineJob *jobs, const int2 *lists,
                                                                  const int4 *meta, int2 *bridges, int4 *bmeta,
                                                                  unsigned *status, unsigned long long *work, int n_jobs,
                                                                  const int *ext_list = nullptr, int2 *ext = nullptr, int *ext_slot = nullptr,
                                                                  int slot_base = 0, int budget = BR_MAX, int ext_stride = EXT_MAX,
                                                                  LatHelp lat = LAT_NONE)
{
    __shared__ unsigned long long lat_e;               // a published chunk result / a decision of thread 0 (broadcast)
    __shared__ unsigned long long lat_state[EXT ? 1 : LAT_D];
    __shared__ int lat_prog[EXT ? 1 : LAT_D];
    __shared__ SharedT<64> shw[BR_LA];                 // one scratch per wave
    __shared__ int lst[LST_MAX];                       // downstream anchor positions (shared by the waves)
    __shared__ int2 obuf[BR_MAX];
    __shared__ int2 res[BR_LA];                        // per wave: (outcome, value)
    if ((*status & ST_WIDE_RANGE) != 0u) return;       // K0 refused the data: the host redoes the call
    // (the single-wave bridge kernel's hint, requested here so that nobody waits for it at the end: do idle workgroups stay to help?)
    const bool lat_stay = !EXT && lat.res != nullptr && c.half > 0 && (lat.ctl[0] != 0ull || lat.stay != 0);
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    SharedT<64> &sh = shw[wave];
    unsigned bad = 0;
    Work wk = PS_WORK_INIT;
    enum : int { O_CONT = 0, O_HIT = 1, O_EARLY = 2, O_LATE = 3, O_NONE = 4 };
    const int LIMIT = EXT ? BR_MAX + ext_stride : budget;
    for (int gi = blockIdx.x; gi < n_jobs; gi += gridDim.x) {
        const int g = EXT ? ext_list[gi] : gi;
        const int4 bm = bmeta[g];
        if (!EXT && bm.w != BR_DEFER) continue;        // only the seams the single-wave bridge kernel deferred
        const SpineJob job = jobs[g];
        const int4 m = meta[g];
        int cnt = bm.x;                                // anchors the seam already has; the chain resumes behind the last one
        for (int i = threadIdx.x; i < cnt && i < BR_MAX; i += 64 * BR_LA) obuf[i] = bridges[static_cast<int64_t>(g) * BR_MAX + i];
        __syncthreads();
        const long long slot = slot_base + gi;         // (EXT) this seam's part of the side buffer
        int a = cnt > 0 ? obuf[cnt - 1].x : (m.x > 0 ? lists[job.out_off + m.x - 1].x : job.start);   // (EXT: cnt <= BR_MAX, the host sees to it)
        int st = BR_FAIL, jt = -1, ji = 0, cached = -1, ccnt = 0;
        const EvRef er = ev_ref_of(c, job.ev);
        // window of the current find_split to resume at (EXT: an open tile resumes behind the windows of its own chain)
        long long jres = EXT ? (cnt == 0 && m.x == 0 ? m.z : 0) : bm.y;
        // the seam's listed stretch, if any (see LAT_D): slot, lattice origin, tag
        const bool lat_use = lat_stay;                 // (no helpers, no listing)
        int lslot = -1;
        long long la0 = 0;
        unsigned ltag = 0;
        bool lat_valid = false;
        for (int step = cnt; step <= LIMIT; ++step) {
            int u = a / job.tile_len;
            if (u > job.ntiles - 1) u = job.ntiles - 1;
            u += job.first_tile;
            const SpineJob uj = jobs[u];
            if (u != cached) {                         // cache the downstream list's positions in LDS
                ccnt = meta[u].x;
                __syncthreads();
                for (int i = threadIdx.x; i < ccnt && i < LST_MAX; i += 64 * BR_LA) lst[i] = lists[uj.out_off + i].x;
                __syncthreads();
                cached = u;
            }
            int found = -2;                            // -1: a is the tile start, >=0: index in its list
            if (a == uj.start) found = -1;
            else {
                int lo = 0, hi = ccnt - 1;
                while (lo <= hi) {
                    const int mid = (lo + hi) >> 1;
                    const int v = mid < LST_MAX ? lst[mid] : lists[uj.out_off + mid].x;
                    if (v == a) { found = mid; break; }
                    if (v < a) lo = mid + 1; else hi = mid - 1;
                }
            }
            if (found != -2 && u != g) { st = BR_JOINED; jt = u; ji = found; break; }
            if (step == LIMIT) break;
            // find_split(a, job.end) with look-ahead
            const long long start = a, end = job.end;
            const long long lim = end - 2LL * c.mw;
            int kind = KIND_NONE, s = -1;
            const long long jfirst = jres;
            jres = 0;
            // Window j of this find_split is lattice window Jbase + j: of the listed lattice while the phase holds (the find_split
            // that listed it, and those behind forced splits), else of the find_split's own (Jbase 0), not listed yet.  Steps are
            // aligned to multiples of four lattice windows so that chunk boundaries are step boundaries.
            bool on_lat = lat_valid && start >= la0 && (start - la0) % c.half == 0;
            if (!on_lat && lat_valid) {                                        // a real split broke the phase: the listed stretch is over
                lat_valid = false;
                if (threadIdx.x == 0) lat_st(&lat.prog[lslot], 0);
            }
            long long Jbase = on_lat ? (start - la0) / c.half : 0;
            int nw = BR_LA;
            for (long long j = jfirst;; j += nw) {
                nw = (lat_use && Jbase + j >= LAT_W / 2) ? BR_LA - static_cast<int>((Jbase + j) & (BR_LA - 1)) : BR_LA;   // (aligned from the eighth window on)
                if (lat_use && lslot != -2 && ((Jbase + j) & (LAT_W - 1)) == 0 && Jbase + j >= LAT_W) {
                    // ---- a chunk boundary, LAT_W or more windows into a stretch ----
                    const long long ci = (Jbase + j) / LAT_W;
                    if (!on_lat) {                                             // a stretch on a lattice of its own: list it
                        if (threadIdx.x == 0) {
                            int sl = lslot;
                            if (sl < 0) {
                                const unsigned long long t = atomicAdd(&lat.ctl[1], 1ull);
                                sl = t < static_cast<unsigned long long>(LAT_D) ? static_cast<int>(t) : -2;
                                if (sl >= 0) lat_st(&lat.seam[sl], g);
                            }
                            unsigned long long tg = 0;
                            if (sl >= 0) {
                                tg = atomicAdd(&lat.ctl[2], 1ull);
                                if (tg >= static_cast<unsigned long long>(LAT_TAGS)) sl = -2;
                            }
                            lat_e = (static_cast<unsigned long long>(static_cast<unsigned>(sl)) << 32) | (lat.tag_base + static_cast<unsigned>(tg));
                        }
                        __syncthreads();
                        lslot = static_cast<int>(static_cast<unsigned>(lat_e >> 32));
                        ltag = static_cast<unsigned>(lat_e);
                        __syncthreads();
                        if (lslot >= 0 && ci < 256) {
                            la0 = start; Jbase = 0; lat_valid = true; on_lat = true;
                            if (threadIdx.x == 0)
                                lat_st(&lat.state[lslot], static_cast<unsigned long long>(static_cast<unsigned>(start)) |
                                                          (static_cast<unsigned long long>(ltag) << 32) | (static_cast<unsigned long long>(ci) << 56));
                        }
                    }
                    if (on_lat && ci < LAT_C) {
                        if (threadIdx.x == 0) {
                            // where the owner is (release: the slot's seam and state are out before a helper sees the slot active)
                            __hip_atomic_store(&lat.prog[lslot], static_cast<int>(ci) + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                            lat_e = lat_ld(&lat.res[static_cast<long long>(lslot) * LAT_C + ci]);
                        }
                        __syncthreads();
                        const unsigned long long e = lat_e;
                        __syncthreads();
                        if (static_cast<unsigned>(e >> 40) == ltag) {                          // a helper has been here
                            const int pos = static_cast<int>(static_cast<unsigned>(e));
                            const int off = static_cast<int>((e >> 32) & 15u);
                            const long long last = pos >= 0 ? j + off : j + LAT_W - 1;          // the last window the result stands for
                            const long long ps_last = start + last * c.half;
                            if (ps_last < lim && ps_last <= start + c.maxw) {                   // (none of them ends the loop or forces a split)
                                if (threadIdx.x == 0) atomicAdd(&lat.ctl[5], 1ull);               // (counter: a published chunk taken)
                                if (pos >= 0) { kind = KIND_HIT; s = pos; break; }
                                nw = LAT_W;
                                continue;
                            }
                        }
                    }
                }
                const long long ps = start + (j + wave) * c.half;
                int oc = O_CONT, val = -1;
                if (wave >= nw) {                                              // (alignment step: this wave sits it out)
                } else if (ps >= lim) {                                        // the loop is over: :199-201
                    if (end - start <= c.maxw) oc = O_NONE;
                    else { oc = O_LATE; const long long x = start + c.maxw, y = end - c.mw; val = static_cast<int>(x < y ? x : y); }
                } else if (ps > start + c.maxw) {                              // :189-191
                    oc = O_EARLY; const long long x = start + c.maxw, y = end - c.mw; val = static_cast<int>(x < y ? x : y);
                } else {
                    long long pe = ps + c.W;
                    if (pe > end) pe = end;
                    if (pe - ps > 2LL * c.mw) {
                        wk.windows += 1; wk.cands += pe - ps - 2LL * c.mw + 1;
                        val = scan_window_ph<DT>(c, er, job.base, static_cast<int>(ps), static_cast<int>(pe),
                                                 static_cast<int>(ps) + c.mw, static_cast<int>(pe) - c.mw, c.min_gain, sh, bad, wk);
                        if (val >= 0) oc = O_HIT;
                    }
                }
                if ((threadIdx.x & 63) == 0) res[wave] = make_int2(oc, val);
                __syncthreads();
                int first = -1;
#pragma unroll
                for (int w = BR_LA - 1; w >= 0; --w) if (res[w].x != O_CONT) first = w;
                const int2 r = res[first < 0 ? 0 : first];
                __syncthreads();
                if (first >= 0) {
                    kind = r.x == O_HIT ? KIND_HIT : r.x == O_EARLY ? KIND_EARLY : r.x == O_LATE ? KIND_LATE : KIND_NONE;
                    s = r.y;
                    break;
                }
            }
            if (kind == KIND_NONE) { st = BR_ENDED; break; }
            if (threadIdx.x == 0) {
                if (!EXT || cnt < BR_MAX) obuf[cnt] = make_int2(s, kind);
                else ext[slot * ext_stride + (cnt - BR_MAX)] = make_int2(s, kind);
            }
            ++cnt;
            a = s;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < cnt && i < BR_MAX; i += 64 * BR_LA) bridges[static_cast<int64_t>(g) * BR_MAX + i] = obuf[i];
        if (threadIdx.x == 0) {
            bmeta[g] = make_int4(cnt, jt, ji, st);
            if (EXT) ext_slot[g] = static_cast<int>(slot);
            if (lslot >= 0) lat_st(&lat.prog[lslot], 0);                       // nothing left to help with
        }
        __syncthreads();                               // obuf / lst are reused by the next tile
    }
    // ---- helper: chunks of the listed stretches, ahead of their owners (see LAT_D) -----------------------------------------
    if constexpr (!EXT) {
        if (lat_stay) {
            if (threadIdx.x == 0) atomicAdd(&lat.ctl[3], 1ull);                // through with my own seams
            const bool stay = true;
            int idle_polls = 0;
            const long long t_start = wall_clock64();
            const int H = static_cast<int>(gridDim.x);
            while (stay) {
                if (threadIdx.x == 0) lat_e = (min(lat_ld(&lat.ctl[1]), static_cast<unsigned long long>(LAT_D)) << 32) | lat_ld(&lat.ctl[3]);
                __syncthreads();
                const int D = static_cast<int>(lat_e >> 32);
                const bool all_through = static_cast<unsigned>(lat_e) >= gridDim.x;
                __syncthreads();
                for (int t = threadIdx.x; t < D; t += 64 * BR_LA) {
                    lat_prog[t] = __hip_atomic_load(&lat.prog[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - 1;   // (-1: not active)
                    lat_state[t] = lat_ld(&lat.state[t]);
                }
                int act = 0;
                __syncthreads();
                for (int t = threadIdx.x; t < D; t += 64 * BR_LA) act |= lat_prog[t] >= 0;
                const bool any_active = __syncthreads_or(act) != 0;
                // Out when nothing is listed and either every workgroup of the launch is through with its own seams, or nothing
                // has been listed for a while (~0.1 ms): workgroups of this launch that have not started yet may be waiting for
                // the very slots the helpers sit on -- a helper must never wait for them.
                idle_polls = any_active ? 0 : idle_polls + 1;
                if (!any_active && (all_through || (idle_polls > 32 && !lat.stay))) break;
                if (wall_clock64() - t_start > 5000000LL) break;              // (50 ms at 100 MHz: a guard, never the way out)
                bool worked = false;
                for (int sI = 0; sI < D; ++sI) {
                    const int pr = lat_prog[sI];
                    if (pr < 0) continue;
                    const unsigned long long stw = lat_state[sI];
                    const int cfirst = static_cast<int>(stw >> 56);
                    const unsigned tag = static_cast<unsigned>(stw >> 32) & 0xffffffu;
                    const long long a0 = static_cast<long long>(static_cast<unsigned>(stw));
                    const int look = max(4, min(H, 2 * (pr - cfirst) + 4));   // how far ahead of the owner: grows with its progress
                    const int r = (static_cast<int>(blockIdx.x) + H - (sI * 7) % H) % H;
                    const int ci = pr + 1 + ((r - (pr + 1) % H + H) % H);     // my chunk among the next H
                    if (ci > pr + look || ci >= LAT_C) continue;
                    unsigned long long *slot_p = &lat.res[static_cast<long long>(sI) * LAT_C + ci];
                    if (threadIdx.x == 0) lat_e = (static_cast<unsigned>(lat_ld(slot_p) >> 40) == tag) ? 1ull : 0ull;
                    __syncthreads();
                    const bool published = lat_e != 0ull;
                    __syncthreads();
                    if (published) continue;
                    const int hg = lat_ld(&lat.seam[sI]);
                    if (hg < 0 || hg >= n_jobs) continue;                      // (cannot be: the slot's seam is out before the slot is active)
                    const SpineJob job = jobs[hg];
                    const long long end = job.end, lim = end - 2LL * c.mw;
                    const long long ps0 = a0 + static_cast<long long>(ci) * LAT_W * c.half;
                    const EvRef er = ev_ref_of(c, job.ev);
                    int hit_off = 0, hit_pos = -1;
                    for (int rr = 0; rr < LAT_W / BR_LA; ++rr) {
                        const long long ps = ps0 + static_cast<long long>(rr * BR_LA + wave) * c.half;
                        int val = -1;
                        if (ps < lim) {
                            long long pe = ps + c.W;
                            if (pe > end) pe = end;
                            if (pe - ps > 2LL * c.mw) {
                                wk.windows += 1; wk.cands += pe - ps - 2LL * c.mw + 1;
                                val = scan_window_ph<DT>(c, er, job.base, static_cast<int>(ps), static_cast<int>(pe),
                                                         static_cast<int>(ps) + c.mw, static_cast<int>(pe) - c.mw, c.min_gain, sh, bad, wk);
                            }
                        }
                        if ((threadIdx.x & 63) == 0) res[wave] = make_int2(val >= 0 ? O_HIT : O_CONT, val);
                        __syncthreads();
                        int first = -1;
#pragma unroll
                        for (int w = BR_LA - 1; w >= 0; --w) if (res[w].x != O_CONT) first = w;
                        const int2 rv = res[first < 0 ? 0 : first];
                        __syncthreads();
                        if (first >= 0) { hit_off = rr * BR_LA + first; hit_pos = rv.y; break; }
                    }
                    if (threadIdx.x == 0) {
                        lat_st(slot_p, (static_cast<unsigned long long>(tag) << 40) | (static_cast<unsigned long long>(hit_off & 15) << 32) | static_cast<unsigned>(hit_pos));
                        atomicAdd(&lat.ctl[4], 1ull);                                              // (counter: a chunk result published)
                    }
                    worked = true;
                }
                if (!worked) __builtin_amdgcn_s_sleep(32);
                __syncthreads();                                               // lat_prog / lat_state are rewritten by the next round
            }
        }
    }
    flush_wave(bad, wk, status, work, 1);              // counters: every wave counted its own scans
}

#ifndef PS_TREE_ROWSKIP
#define PS_TREE_ROWSKIP false
#endif
// ---- phase 3: in-order traversal of rec(start, end) -----------------------------------------------
// One job, by the NT threads that share `sh` (a workgroup, or one wave of a multi-wave workgroup when NT == 64).
template <int NT, int DT, bool BSONLY = false>
__device__ __forceinline__ int tree_job(const DevCfg &c, int *ys, const TreeJob &job, long long ji, int32_t *scratch,
                                        int2 *spill, int32_t *counts, SharedT<NT> &sh, unsigned &bad, Work &wk, int32_t *blk = nullptr)
{
    const int tid = ps_tid<NT>();
    int32_t *out = scratch + job.out_off;
    int2 *sp_glob = spill + job.out_off;
    int start = job.start, end = job.end, j0 = job.j0, sp = 0, cnt = 0, flushed = 0;
    const EvRef er = {job.m, job.boff, job.pad_};       // (pad_: the event's phase, assemble_items_kernel)
    int *obuf = reinterpret_cast<int *>(sh.obuf);
    constexpr int OB = 2 * SharedT<NT>::OB;
    auto emit = [&](int v) {
        if (cnt - flushed == OB) {                     // rare: spill the LDS buffer to the private scratch
            ps_sync<NT>();
            for (int i = tid; i < OB; i += NT)
                if (flushed + i < job.out_cap) out[flushed + i] = obuf[i];
            flushed += OB;
            ps_sync<NT>();
        }
        if (cnt >= job.out_cap) bad |= ST_OUT_OVERFLOW;
        if (tid == 0) obuf[cnt - flushed] = v;
        ++cnt;
    };
    for (;;) {
        int kind;
        // (no row skipping in subtree windows: they rarely hold a split, the extra registers cost 2-3 %; filtered events on
        //  the 64-bit digest, where they mostly do, neither gain nor lose -- measured)
        int s = find_split<NT, DT, false, BSONLY, PS_TREE_ROWSKIP>(c, ys, job.base, start, end, j0, kind, sh, bad, wk, job.end, er);
        if (kind == KIND_NONE) {
            if (sp == 0) break;
            --sp;
            if (tid == 0) sh.pop = sp < SharedT<NT>::SN ? sh.stack[sp] : sp_glob[sp - SharedT<NT>::SN];
            ps_sync<NT>();
            int2 top = sh.pop;
            if constexpr (NT == 64) { top.x = __builtin_amdgcn_readfirstlane(top.x); top.y = __builtin_amdgcn_readfirstlane(top.y); }   // (uniform: scalar registers)
            ps_sync<NT>();
            emit(top.x);
            start = top.x; end = top.y; j0 = 0;
            continue;
        }
        if (kind == KIND_EARLY) {                     // [split] + rec(split, end)
            emit(s);
            start = s; j0 = 0;
            continue;
        }
        // HIT / LATE: rec(start, s) first, then emit s and continue with rec(s, end)
        if (sp < SharedT<NT>::SN) {
            if (tid == 0) sh.stack[sp] = make_int2(s, end);
        } else if (sp - SharedT<NT>::SN < job.out_cap) {
            if (tid == 0) sp_glob[sp - SharedT<NT>::SN] = make_int2(s, end);
        } else {
            bad |= ST_STACK_OVERFLOW;
            break;
        }
        ++sp;
        ps_sync<NT>();
        j0 = left_child_j0(start, s, c.W, c.half);
        end = s;
    }
    if (cnt > job.out_cap) cnt = job.out_cap;
    ps_sync<NT>();
    for (int i = flushed + tid; i < cnt; i += NT) out[i] = obuf[i - flushed];
    // (the counts are zero before the kernel runs -- assemble_items_kernel, or a memset on the host-stitch path -- and
    //  most jobs find nothing: no store then, and nothing for the fence below to wait for -- a store's round trip at the
    //  end of every job is 1-2 us in front of the next job's loads)
    // (blk: the sum of the counts per 256 jobs -- what gather_scan_kernel needs to place a block of items without a scan kernel
    //  in front of it; a few hundred atomics per call, on jobs that found something)
    if (tid == 0 && counts && cnt) { counts[ji] = cnt; if (blk) atomicAdd(&blk[ji >> GS_LOG], cnt); }
    ps_sync<NT>();                                     // obuf / stack are reused by the next job
    return cnt;
}

template <int NT, int DT>
__global__ __launch_bounds__(NT, (NT == 64 ? PS_BS_MINW : 4)) PS_SCAN_REGS void tree_kernel(DevCfg c, const TreeJob *__restrict__ jobs, int32_t *scratch,
                                                  int2 *spill, int32_t *counts, unsigned *status,
                                                  unsigned long long *work, long long n_jobs_host, const AsmHeader *hdr,
                                                  int jobs_per_wave, int par_max_jobs = 0)
{
    extern __shared__ int ys[];
    __shared__ SharedT<NT> sh;
    const long long n_jobs = dev_count(hdr, n_jobs_host);
    if (n_jobs <= par_max_jobs) return;                // (few, deep jobs: tree_par_kernel, launched next to this one, takes the call)
    // How many of the launched slots work (jobs_per_wave > 0; the job count is only known on the device).  With few
    // jobs per slot the kernel's duration is a handful of windows in a row, and a window's latency doubles from two to
    // four waves per SIMD: then half the slots do the work -- and the other half of the SIMD's registers is where
    // another call's kernels run meanwhile.  With many jobs per slot the SIMD's throughput counts, which is 1.4 x
    // higher with four waves than with two.  Between: jobs / jobs_per_wave slots.  (bench trace, 9 231 jobs: 2 307
    // slots; four calls in flight 0.274 -> 0.245 ms per step.  1e9-sample trace, 98 007 jobs: all 4 096, subtree
    // kernel 0.68 ms against 0.90 ms on half of them.)
    long long G = gridDim.x;
    if (jobs_per_wave > 0) {
        G = max(1LL, max(G / 2, min(G, n_jobs / jobs_per_wave)));      // (never 0: a one-workgroup grid with fewer than jobs_per_wave jobs)
        if (static_cast<long long>(blockIdx.x) >= G) return;
    }
    unsigned bad = 0;
    Work wk = PS_WORK_INIT;
    // One workgroup per resident slot, striding over the jobs (a workgroup per job costs more in launches than the
    // one or two scans of a typical job: 0.28 ms against 0.17 ms for the 9 231 jobs of the bench trace).
    for (long long ji = blockIdx.x; ji < n_jobs; ji += G) {
        const TreeJob job = jobs[ji];
        // (all fields in one round trip: otherwise the compiler fetches out_cap, tests it, and only then the rest)
        asm volatile("" : : "s"(job.base), "s"(job.start), "s"(job.end), "s"(job.j0), "s"(job.out_off), "s"(job.m), "s"(job.boff));
        if (job.out_cap == 0) continue;                // spine anchor without a left subtree (device stitch)
        tree_job<NT, DT>(c, ys, job, ji, scratch, spill, counts, sh, bad, wk, hdr ? counts + n_jobs_host : nullptr);
    }
    flush(bad, wk, status, work, 2);
}

// Block-sum scan, TREE_W waves per workgroup: every wave runs single-wave jobs on its own, and the waves of a
// workgroup share the workgroup's strided list of jobs through a counter in LDS -- a wave that got short jobs takes
// more of them (bench trace: 0.165 -> 0.146 ms; 4 waves: 0.150).  A counter in HBM for all workgroups costs more than
// it balances: the returning atomic sits in front of the scan's loads in the wave's in-order memory counter.
#ifndef PS_TREE_W
#define PS_TREE_W 2           // waves per workgroup.  8 (a whole CU at two waves per SIMD) is the fastest for one call at a
#endif                        // time (0.145 ms against 0.150 with 2 and 0.170 with single-wave workgroups and fixed shares);
                              // with several calls in flight a workgroup of 8 waves needs an entirely empty CU before it can
                              // start under another call's spine kernel, 2-wave workgroups fill slots as they free up
                              // (4 streams: 0.354 -> 0.347 ms per step; single-wave workgroups 0.345)
constexpr int TREE_W = PS_TREE_W;
template <int DT>
__global__ __launch_bounds__(64 * TREE_W, PS_BS_MINW) PS_SCAN_REGS void tree_mw_kernel(DevCfg c, const TreeJob *__restrict__ jobs, int32_t *scratch,
                                                  int2 *spill, int32_t *counts, unsigned *status,
                                                  unsigned long long *work, long long n_jobs_host, const AsmHeader *hdr,
                                                  unsigned long long *tail_ctr, int tail_pct)
{
    extern __shared__ int ys[];                        // TREE_W x SharedT<64> (beyond the static 64 KB for 8 waves)
    __shared__ int next_k;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));     // (scalar: the wave's LDS block gets a scalar base)
    SharedT<64> &sh = reinterpret_cast<SharedT<64> *>(ys)[wave];
    const long long n_jobs = dev_count(hdr, n_jobs_host);
    // tail_pct > 0 (option tree_tail_pct, default 0): the last tail_pct per cent of the job list are drawn one by one
    // from a counter in HBM by whoever has finished its static share.  Measured slower at every share (15 %: 0.146 ->
    // 0.160 ms): the returning atomic sits in front of the job's first loads, as with a fully dynamic draw.
    const long long n_static = n_jobs - n_jobs * tail_pct / 100;
    unsigned bad = 0;
    Work wk = PS_WORK_INIT;
    if (threadIdx.x == 0) next_k = TREE_W;             // the first TREE_W jobs of the list are the waves' own
    __syncthreads();
    bool dyn = false;                                  // (one loop, one call site of the job body: a second inlined copy
    for (long long k = wave;;) {                       //  of the scan code costs registers -- 0.146 -> 0.204 ms)
        long long ji = blockIdx.x + k * gridDim.x;
        if (!dyn && ji >= n_static) dyn = true;
        if (dyn) {
            if (n_static >= n_jobs) break;
            unsigned long long t = 0;
            if (ps_tid<64>() == 0) t = atomicAdd(tail_ctr, 1ULL);
            ji = n_static + static_cast<long long>(__shfl(t, 0));
            if (ji >= n_jobs) break;
        }
        const TreeJob job = jobs[ji];
        if (job.out_cap != 0) tree_job<64, DT>(c, nullptr, job, ji, scratch, spill, counts, sh, bad, wk, hdr ? counts + n_jobs_host : nullptr);
        if (!dyn) {
            int kk = 0;
            if (ps_tid<64>() == 0) kk = atomicAdd(&next_k, 1);
            k = __builtin_amdgcn_readfirstlane(kk);
        }
    }
    flush_wave(bad, wk, status, work, 2);
}

// ---- subtree jobs whose recursion is DEEP: the waves of a workgroup share ONE job ---------------------------------------
// rec(start, s) and rec(s, end) (cparsers.pyx:203) are independent of each other: after a split the left part is somebody
// else's work.  A filtered event (the reference's default workflow, DataTypes.py:975-984) has a few hundred subtree jobs
// of ~60 window scans each -- on one wave per job nine tenths of the chip idle while every job walks its recursion depth
// first.  Here the PAR_W waves of a workgroup take one job together: an interval queue in LDS (append-only, tickets; no
// traffic leaves the CU -- round 3 tried a queue in HBM for all workgroups and lost to the contention on its lines), the
// wave that finds a split appends the LEFT part and goes on with the right one, a wave without work takes the next
// ticket and waits for its slot to be filled or for the count of unfinished intervals to reach zero.  Boundaries are
// collected unordered in LDS and written by rank.  The scans and their order within an interval are tree_job's: the same
// windows, the same decisions.  Jobs too large for the LDS queue (more than PAR_ON possible boundaries) are walked by
// wave 0 alone, as in tree_kernel.
#ifndef PS_PAR_W
#define PS_PAR_W 4
#endif
constexpr int PAR_W = PS_PAR_W;
constexpr int PAR_ON = 768;                            // boundaries of a job handled in LDS (job length <= PAR_ON * min_width)
constexpr int PAR_QN = PAR_ON + 8;                     // appended intervals: one per split (the right part stays with its wave) + the job itself
struct ParQ {
    int head, tail, pending, ocnt;                     // tickets handed out, intervals appended, intervals unfinished, boundaries found
    int out[PAR_ON];
    int ready[PAR_QN];
    int4 q[PAR_QN];                                    // (start, end, first window j0, -)
};
template <int DT>
// (its LDS -- PAR_W scan scratches and the queue, 58 KB -- admits two workgroups per CU, i.e. two waves per SIMD: the
//  scan body may keep everything in registers here)
__global__ __launch_bounds__(64 * PAR_W, 2) void tree_par_kernel(DevCfg c, const TreeJob *__restrict__ jobs, int32_t *scratch,
                                                  int2 *spill, int32_t *counts, unsigned *status,
                                                  unsigned long long *work, long long n_jobs_host, const AsmHeader *hdr,
                                                  int par_max_jobs)
{
    extern __shared__ int ys[];                        // PAR_W x SharedT<64>, then the queue
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    SharedT<64> &sh = reinterpret_cast<SharedT<64> *>(ys)[wave];
    ParQ &Q = *reinterpret_cast<ParQ *>(reinterpret_cast<SharedT<64> *>(ys) + PAR_W);
    const long long n_jobs = dev_count(hdr, n_jobs_host);
    // The job count is only known on the device.  With more jobs than workgroup slots one wave per job at four waves per
    // SIMD has the higher throughput (32 filtered events, 7 232 jobs: 1.24 ms against 2.21 ms here): tree_kernel, launched
    // behind this kernel with the same limit, takes those calls and this one leaves.
    if (n_jobs > par_max_jobs) return;
    unsigned bad = 0;
    Work wk = PS_WORK_INIT;
    auto ld = [](const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    for (long long ji = blockIdx.x; ji < n_jobs; ji += gridDim.x) {
        const TreeJob job = jobs[ji];
        if (job.out_cap == 0) continue;                // spine anchor without a left subtree (uniform over the workgroup)
        if (job.out_cap > PAR_ON) {                    // too long for the LDS queue: depth first on one wave
            if (wave == 0) tree_job<64, DT>(c, nullptr, job, ji, scratch, spill, counts, sh, bad, wk);
            __syncthreads();
            continue;
        }
        if (threadIdx.x == 0) {
            Q.head = 0; Q.tail = 1; Q.pending = 1; Q.ocnt = 0;
            Q.q[0] = make_int4(job.start, job.end, job.j0, 0);
            Q.ready[0] = 1;
        }
        for (int i = 1 + threadIdx.x; i < PAR_QN; i += 64 * PAR_W) Q.ready[i] = 0;
        __syncthreads();
        const EvRef er = {job.m, job.boff, job.pad_};
        for (;;) {
            // a ticket, then its interval -- or the end of the job
            int t = 0;
            if (lane == 0) t = atomicAdd(&Q.head, 1);
            t = __builtin_amdgcn_readfirstlane(t);
            bool have = false;
            for (;;) {
                const int r = t < PAR_QN ? ld(&Q.ready[t]) : 0;
                const int p = ld(&Q.pending);          // (read after `ready`: a slot filled later keeps pending above zero)
                if (__builtin_amdgcn_readfirstlane(r)) { have = true; break; }
                if (__builtin_amdgcn_readfirstlane(p) == 0) break;
                __builtin_amdgcn_s_sleep(4);
            }
            if (!have) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const int4 it = Q.q[t];
            int start = __builtin_amdgcn_readfirstlane(it.x), end = __builtin_amdgcn_readfirstlane(it.y), j0 = __builtin_amdgcn_readfirstlane(it.z);
            for (;;) {                                 // this interval, then the right parts of its splits
                int kind;
                const int sp = find_split<64, DT, false, true, PS_TREE_ROWSKIP>(c, nullptr, job.base, start, end, j0, kind, sh, bad, wk, job.end, er);
                if (kind == KIND_NONE) {
                    if (lane == 0) atomicSub(&Q.pending, 1);
                    break;
                }
                if (lane == 0) {
                    const int o = atomicAdd(&Q.ocnt, 1);
                    if (o < PAR_ON) Q.out[o] = sp;
                    if (kind != KIND_EARLY) {          // HIT / LATE: rec(start, sp) goes to the queue (:203); EARLY has no left part (:189-191)
                        // (the slot first: an interval that found no slot -- impossible while boundaries are min_width apart,
                        //  out_cap bounds their number -- must not be counted as unfinished, or nobody would ever see zero;
                        //  this wave's own interval keeps `pending` above zero meanwhile)
                        const int i = atomicAdd(&Q.tail, 1);
                        if (i < PAR_QN) {
                            atomicAdd(&Q.pending, 1);
                            Q.q[i] = make_int4(start, sp, left_child_j0(start, sp, c.W, c.half), 0);
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            __hip_atomic_store(&Q.ready[i], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
                start = sp; j0 = 0;                    // rec(sp, end)
            }
        }
        __syncthreads();
        // boundaries by rank (they are distinct), then the count
        const int cnt = Q.ocnt;
        if (cnt > PAR_ON || cnt > job.out_cap || Q.tail > PAR_QN) bad |= ST_OUT_OVERFLOW;
        else {
            int32_t *out = scratch + job.out_off;
            for (int i = threadIdx.x; i < cnt; i += 64 * PAR_W) {
                const int v = Q.out[i];
                int rank = 0;
                for (int k = 0; k < cnt; ++k) rank += Q.out[k] < v ? 1 : 0;
                out[rank] = v;
            }
            if (threadIdx.x == 0 && counts && cnt) { counts[ji] = cnt; if (hdr) atomicAdd(&counts[n_jobs_host + (ji >> GS_LOG)], cnt); }
        }
        __syncthreads();                               // the queue is reused by the next job
    }
    flush_wave(bad, wk, status, work, 2);
}

// ---- single scans for the API-completeness entry points -----------------------------------------
// mode 0: score_samples(no_split=True)  (window [0,n), candidates mw..n-mw, threshold min_gain)
// mode 1: best_single_split              (window [0,n-1), candidates 2..n-4, threshold 0)
template <int NT, int DT>
__global__ __launch_bounds__(NT) void single_scan_kernel(DevCfg c, int n, int mode, double *scores,
                                                         double *gain_out, int *idx_out,
                                                         unsigned *status, unsigned long long *work)
{
    extern __shared__ int ys[];
    __shared__ SharedT<NT> sh;
    unsigned bad = 0;
    Work wk = PS_WORK_INIT;
    int r = -1;
    double g = mode == 0 ? c.min_gain : 0.0;
    if (mode == 0) {
        if (n > 2 * c.mw)
            r = scan_window<NT, DT, true>(c, ys, 0, 0, n, c.mw, n - c.mw, c.min_gain, scores, sh, bad, wk, &g);
    } else {
        const int end = n - 1;
        if (end >= 1) r = scan_window<NT, DT, true>(c, ys, 0, 0, end, 2, end - 3, 0.0, nullptr, sh, bad, wk, &g);
    }
    if (threadIdx.x == 0) { *idx_out = r; *gain_out = g; }
    flush(bad, wk, status, work);
}

// ---- gather: final boundary list = for every true spine anchor: [its left subtree..., anchor] ----
struct Item { int32_t job; int32_t anchor; };

// single-workgroup exclusive scan of (tree count + 1) per item -> pos[n_items + 1]; then the per-event
// offsets bounds_off[e] = pos[first_item[e]].
// 16 384 items per trip (one trip for the bench trace).  Loads and stores are coalesced (item b0 + 1024 k + t by thread
// t) and pass through LDS, where every thread scans 16 CONSECUTIVE values: round 2 gave every thread 16 consecutive
// items in global memory, i.e. 64 cache lines per load instruction and 16 K line requests per trip through one CU's
// L1 (17.8 us for 9 735 items, 0.19 ms for the 98 007 of the 1e9-sample trace).
__global__ __launch_bounds__(1024) void item_scan_kernel(const Item *items, const int32_t *counts,
                                                         int64_t n_items_host, int64_t *pos, const AsmHeader *hdr,
                                                         const int64_t *first_item, int32_t n_ev, int64_t *bounds_off)
{
    constexpr int PER = 16;
    constexpr int TRIP = 1024 * PER;
    if (hdr && hdr->fail) {                            // failed / refused stitch: first_item is not valid either
        for (int e = threadIdx.x; e <= n_ev; e += 1024) bounds_off[e] = 0;
        if (threadIdx.x == 0) pos[0] = 0;
        return;
    }
    const int64_t n_items = dev_count(hdr, n_items_host);
    __shared__ int vals[TRIP];                         // the trip's values, then their exclusive prefix (relative to the trip)
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t b0 = 0; b0 < n_items; b0 += TRIP) {
        // 1. coalesced: value of item b0 + 1024 k + t
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int64_t i = b0 + 1024 * k + threadIdx.x;
            int v = 0;
            if (i < n_items) {
                const int jb = items[i].job;
                // (the device stitch numbers jobs like items: usually the coalesced load is the right one)
                v = 1 + (jb < 0 ? 0 : jb == i ? counts[i] : counts[jb]);
            }
            vals[1024 * k + threadIdx.x] = v;
        }
        __syncthreads();
        // 2. every thread: 16 consecutive values
        int v[PER];
        long long tot = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) { v[k] = vals[PER * threadIdx.x + k]; tot += v[k]; }
        long long inc = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            long long u = __shfl_up(inc, d);
            if (lane >= d) inc += u;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        long long run = inc - tot;                     // relative to the trip
        for (int w = 0; w < wave; ++w) run += wsum[w];
        const long long base = carry_s;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            vals[PER * threadIdx.x + k] = static_cast<int>(run);      // (a trip holds < 2^31 boundaries)
            run += v[k];
        }
        __syncthreads();
        // 3. coalesced stores
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int64_t i = b0 + 1024 * k + threadIdx.x;
            if (i < n_items) pos[i] = base + vals[1024 * k + threadIdx.x];
        }
        if (threadIdx.x == 1023) carry_s = base + run;
        __syncthreads();
    }
    if (threadIdx.x == 0) pos[n_items] = carry_s;
    __syncthreads();                                   // pos[] written above is read by other threads below
    for (int e = threadIdx.x; e <= n_ev; e += 1024) bounds_off[e] = pos[first_item[e]];
}

__global__ __launch_bounds__(64) void gather_kernel(const Item *items, const TreeJob *jobs,
                                                    const int32_t *counts, const int32_t *scratch,
                                                    const int64_t *pos, int64_t n_items_host,
                                                    int32_t *bounds, int64_t cap, uint8_t *is_spine, const AsmHeader *hdr)
{
    const int64_t n_items = dev_count(hdr, n_items_host);
    for (int64_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const Item item = items[it];
        const int64_t p = pos[it];
        int cnt = 0;
        if (item.job >= 0) {
            cnt = counts[item.job];
            const int32_t *src = scratch + jobs[item.job].out_off;
            for (int i = threadIdx.x; i < cnt; i += 64)
                if (p + i < cap) { bounds[p + i] = src[i]; if (is_spine) is_spine[p + i] = 0; }
        }
        if (threadIdx.x == 0 && p + cnt < cap) { bounds[p + cnt] = item.anchor; if (is_spine) is_spine[p + cnt] = 1; }
    }
}



// ---- gather fused with its scan (round 6; device-stitch path: job index == item index) ------------------------------
// The position of item i in the result is i + (sum of the subtree counts before it).  The subtree kernels leave the sum of the
// counts per GS = 256 jobs in blk[] (a few hundred atomics per call: most jobs find nothing), so a workgroup places ITS 256
// items from the block sums before it and a scan of its own 256 values -- no single-workgroup scan kernel between the
// subtrees and the gather (item_scan_kernel waited ~100 us for a CU with sixteen free wave slots when sixteen calls were in
// flight).  The per-event offsets bounds_off[e] = pos[first_item[e]] are written by the workgroup whose block holds
// the event's first item (first_item ascends: two binary searches per block; a first version summed per event -- up to 255
// dependent loads each -- and cost BASELINE config 2's 1 024 events 25 us).
__global__ __launch_bounds__(256) void gather_scan_kernel(const Item *items, const TreeJob *jobs, const int32_t *counts,
                                                          const int32_t *scratch, long long n_items_host, int32_t *bounds,
                                                          int64_t cap, uint8_t *is_spine, const AsmHeader *hdr,
                                                          const int64_t *first_item, int32_t n_ev, int64_t *bounds_off)
{
    constexpr int GS = 1 << GS_LOG;
    __shared__ long long wsum[GS / 64];
    __shared__ int ws[GS / 64];
    __shared__ long long pos_s[GS];
    __shared__ int ev_lo_s, ev_hi_s;
    const bool failed = hdr && hdr->fail;              // failed / refused stitch: first_item is not valid either
    const long long n_items = dev_count(hdr, n_items_host);
    const int32_t *blk = counts + n_items_host;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (failed) {
        for (long long e = blockIdx.x * static_cast<long long>(GS) + tid; e <= n_ev; e += gridDim.x * static_cast<long long>(GS)) bounds_off[e] = 0;
        return;
    }
    // (the block that holds index n_items -- one past the last item -- runs too: events that start there, and the total)
    const long long nb = (n_items >> GS_LOG) + 1;
    for (long long b = blockIdx.x; b < nb; b += gridDim.x) {
        long long part = 0;
        for (long long k = tid; k < b; k += GS) part += blk[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_down(part, d);
        const long long i = (b << GS_LOG) + tid;
        const bool have = i < n_items;
        Item it = {-1, 0};
        int cnt = 0;
        if (have) { it = items[i]; if (it.job >= 0) cnt = counts[it.job]; }
        const int v = have ? cnt + 1 : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(inc, d); if (lane >= d) inc += u; }
        if (lane == 0) wsum[wave] = part;
        if (lane == 63) ws[wave] = inc;
        // the events whose first item lies in this block: first_item ascends with the event index (two binary searches)
        if (tid < 2) {
            const long long key = (b + tid) << GS_LOG;
            int lo = 0, hi = n_ev + 1;                  // first e in [0, n_ev] with first_item[e] >= key
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (first_item[mid] < key) lo = mid + 1; else hi = mid; }
            if (tid == 0) ev_lo_s = lo; else ev_hi_s = lo;
        }
        __syncthreads();
        long long p = (b << GS_LOG) + inc - v;
#pragma unroll
        for (int w = 0; w < GS / 64; ++w) { p += wsum[w]; if (w < wave) p += ws[w]; }
        pos_s[tid] = p;                                // (threads beyond the last item: the position one past it -- their v is 0)
        __syncthreads();
        for (int e = ev_lo_s + tid; e < ev_hi_s; e += GS) bounds_off[e] = pos_s[first_item[e] - (b << GS_LOG)];
        __syncthreads();                               // wsum / ws / pos_s are rewritten by the next block
        const int32_t *src = cnt ? scratch + jobs[it.job].out_off : nullptr;
        // subtrees with many boundaries (filtered events: hundreds): the wave copies them together, one after the other
        unsigned long long big = __ballot(cnt > 16);
        while (big) {
            const int l = __builtin_ctzll(big);
            big &= big - 1ull;
            const long long pl = __shfl(p, l);
            const int cl = __shfl(cnt, l);
            const int32_t *sl = reinterpret_cast<const int32_t *>(__shfl(reinterpret_cast<unsigned long long>(src), l));
            for (int k = lane; k < cl; k += 64)
                if (pl + k < cap) { bounds[pl + k] = sl[k]; if (is_spine) is_spine[pl + k] = 0; }
        }
        if (cnt <= 16)
            for (int k = 0; k < cnt; ++k)
                if (p + k < cap) { bounds[p + k] = src[k]; if (is_spine) is_spine[p + k] = 0; }
        if (have && p + cnt < cap) { bounds[p + cnt] = it.anchor; if (is_spine) is_spine[p + cnt] = 1; }
    }
}

// The call's status block and per-event offsets go back to the host by a KERNEL that writes the pinned buffer (like the
// upload: no hand-over to a copy engine between the last kernel and the host's wake-up).
__global__ __launch_bounds__(256) void download_kernel(const unsigned long long *src, unsigned long long *dst_host, long long n_words)
{
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n_words; i += gridDim.x * 256LL) dst_host[i] = src[i];
}

// ---- device-side stitch: true spine, tree jobs and items from the tile lists and bridges ----------
// exclusive scan of two values over one 1024-thread workgroup chunk with running carries
__device__ __forceinline__ void chunk_exscan2(long long v1, long long v2, long long &e1, long long &e2,
                                              long long *wsum /*[32]*/, long long *carry /*[2]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long i1 = v1, i2 = v2;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long u1 = __shfl_up(i1, d), u2 = __shfl_up(i2, d);
        if (lane >= d) { i1 += u1; i2 += u2; }
    }
    if (lane == 63) { wsum[wave] = i1; wsum[16 + wave] = i2; }
    __syncthreads();
    long long b1 = carry[0], b2 = carry[1];
    for (int w = 0; w < wave; ++w) { b1 += wsum[w]; b2 += wsum[16 + w]; }
    e1 = b1 + i1 - v1; e2 = b2 + i2 - v2;
    __syncthreads();
    if (threadIdx.x == 1023) { carry[0] = b1 + i1; carry[1] = b2 + i2; }
    __syncthreads();
}

// Stitch, part 1 (one workgroup): which tiles lie on the true spine, where the true path enters
// them, how many anchors each contributes.  The per-tile arrays live in LDS when they fit
// (use_lds), otherwise in HBM scratch.
__global__ __launch_bounds__(1024) void assemble_tiles_kernel(
    int n_tiles, const int4 *meta, int4 *bmeta, const int64_t *ev_first_tile, int n_ev,
    int *g_reach, int *g_jump_a, int *g_jump_b, int *entry_out, long long *sp_off_out,
    int64_t *first_item, AsmHeader *hdr, long long max_items, int use_lds, const unsigned *status)
{
    // K0 refused the data (counts too wide for the block sums): the scan kernels did nothing; mark the stitch failed so
    // that everything downstream is a no-op and the host redoes the call with the LDS-window kernels.
    if ((*status & ST_WIDE_RANGE) != 0u) {
        if (threadIdx.x == 0) { hdr->n_items = 0; hdr->n_jobs = 0; hdr->tscratch = 0; hdr->fail = 1; hdr->pad = 0; }
        return;
    }
    extern __shared__ long long dyn_lds[];
    __shared__ long long wsum[32];
    __shared__ long long carry[2];
    __shared__ int fail_s;
    const int tid = threadIdx.x;
    long long *sp_off = use_lds ? dyn_lds : sp_off_out;
    int *lds_i = reinterpret_cast<int *>(dyn_lds + (n_tiles + 1));
    int *reach = use_lds ? lds_i : g_reach;
    int *ja = use_lds ? lds_i + n_tiles : g_jump_a;
    int *jb = use_lds ? lds_i + 2 * n_tiles : g_jump_b;
    int *entry = use_lds ? lds_i + 3 * n_tiles : entry_out;
    if (tid == 0) { fail_s = 0; carry[0] = 0; carry[1] = 0; }
    // A. forward pointers and roots
    for (int g = tid; g < n_tiles; g += 1024) {
        reach[g] = 0;
        entry[g] = 0;
        const int4 b = bmeta[g];
        ja[g] = b.w == BR_JOINED ? b.y : -1;
    }
    __syncthreads();
    for (int e = tid; e < n_ev; e += 1024)
        if (ev_first_tile[e] < ev_first_tile[e + 1]) reach[ev_first_tile[e]] = 1;
    __syncthreads();
    // B. reachability from the roots by pointer doubling: after round k every tile within 2^k
    //    joins of a root is marked
    int rounds = 1;
    while ((1 << rounds) < n_tiles) ++rounds;
    for (int r = 0; r <= rounds; ++r) {
        for (int g = tid; g < n_tiles; g += 1024) {
            const int j = ja[g];
            if (j >= 0 && reach[g]) reach[j] = 1;
            jb[g] = j >= 0 ? ja[j] : -1;
        }
        __syncthreads();
        int *t = ja; ja = jb; jb = t;
    }
    // C. entry index of every reached tile; failed bridges on the path
    for (int g = tid; g < n_tiles; g += 1024) {
        if (!reach[g]) continue;
        const int4 b = bmeta[g];
        if (b.w == BR_JOINED) entry[b.y] = b.z + 1;
        if (b.w == BR_FAIL || b.w == BR_DEFER || b.w == BR_FAIL_REACHED) {
            fail_s = 1;
            if (b.w == BR_FAIL) bmeta[g].w = BR_FAIL_REACHED;      // (for the host: this one lies on the true path)
        }
    }
    __syncthreads();
    // D. contribution of every tile and its offset in the true spine
    for (int g0 = 0; g0 < n_tiles; g0 += 1024) {
        const int g = g0 + tid;
        long long n = 0;
        if (g < n_tiles && reach[g]) {
            const int c = meta[g].x - entry[g];
            n = static_cast<long long>(c) + bmeta[g].x;
            if (c < 0) fail_s = 1;
        }
        long long e1, e2;
        chunk_exscan2(n, 0, e1, e2, wsum, carry);
        if (g < n_tiles) sp_off[g] = e1;
    }
    const long long n_items = carry[0];
    if (tid == 0) sp_off[n_tiles] = n_items;
    __syncthreads();
    for (int e = tid; e <= n_ev; e += 1024) first_item[e] = sp_off[ev_first_tile[e]];
    if (use_lds)
        for (int g = tid; g <= n_tiles; g += 1024) {
            sp_off_out[g] = sp_off[g];
            if (g < n_tiles) entry_out[g] = entry[g];
        }
    if (tid == 0) {
        hdr->n_items = n_items;
        hdr->n_jobs = n_items;
        hdr->tscratch = 0;
        hdr->fail = (fail_s || n_items > max_items) ? 1 : 0;
    }
}

// Stitch, part 2 (grid over the true spine): element i belongs to the tile g with
// sp_off[g] <= i < sp_off[g+1].  Writes the item (anchor) and its tree job: job index = item
// index, output region = (samples of all preceding events + predecessor)/min_width + i, which is
// monotone and non-overlapping without any scan ((a+b)/m >= a/m + b/m for integers).
__global__ __launch_bounds__(256) void assemble_items_kernel(
    const SpineJob *jobs, int n_tiles, const int4 *meta, const int2 *lists, const int2 *bridges,
    const int *entry, const long long *sp_off, long long n_items_host, int mw, int W,
    TreeJob *tjobs, Item *items, int32_t *counts, const AsmHeader *hdr, const int4 *ev_info,
    const int2 *ext = nullptr, const int *ext_slot = nullptr, int ext_stride = EXT_MAX)
{
    const long long n_items = dev_count(hdr, n_items_host);
    // (the sums of the counts per 256 jobs live behind the counts: counts[n_items_host + b], gather_scan_kernel)
    for (long long b = blockIdx.x * 256LL + threadIdx.x; b <= (n_items >> GS_LOG); b += gridDim.x * 256LL) counts[n_items_host + b] = 0;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n_items; i += gridDim.x * 256LL) {
    int lo = 0, hi = n_tiles - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sp_off[mid + 1] > i) hi = mid; else lo = mid + 1;
    }
    const int g = lo;
    const int k = static_cast<int>(i - sp_off[g]);
    const int en = entry[g], cnt = meta[g].x;
    const SpineJob jb2 = jobs[g];
    int2 el;
    int pred;
    if (k < cnt - en) {
        el = lists[jb2.out_off + en + k];
        pred = (en + k) > 0 ? lists[jb2.out_off + en + k - 1].x : jb2.start;
    } else {
        const int kb = k - (cnt - en);
        el = bridge_at(bridges, ext, ext_slot, ext_stride, g, kb);
        // (a tile that gave up before its first anchor is entered at its start: that is the predecessor then)
        pred = kb > 0 ? bridge_at(bridges, ext, ext_slot, ext_stride, g, kb - 1).x : (cnt > 0 ? lists[jb2.out_off + cnt - 1].x : jb2.start);
    }
    const bool has = el.y == KIND_HIT || el.y == KIND_LATE;
    Item it;
    it.anchor = el.x;
    it.job = has ? static_cast<int32_t>(i) : -1;
    items[i] = it;
    TreeJob tj;
    tj.base = jb2.base;
    tj.start = pred;
    tj.end = el.x;
    tj.j0 = left_child_j0(pred, el.x, W, W / 2);
    tj.out_cap = has ? (el.x - pred) / mw + 1 : 0;       // 0 marks "no left subtree": the tree kernel skips it
    tj.out_off = (jb2.vbase + pred) / mw + i;
    tj.pad_ = 0;
    if (ev_info) {
        const int4 info = ev_info[jb2.ev];
        tj.m = info.x;
        tj.pad_ = info.y;                                    // the event's phase (0 unless the digest is trace-aligned)
        tj.boff = (static_cast<long long>(static_cast<unsigned>(info.w)) << 32) | static_cast<unsigned>(info.z);
    } else { tj.m = 0; tj.boff = 0; }
    tjobs[i] = tj;
    counts[i] = 0;
    }
}


// ---- K3: threshold event detector, parsers.py:124-155 ---------------------------------------------
// mask[i] = x[i] < threshold; an edge sits at i (>= 1) when mask[i] != mask[i-1]; pieces lie between
// consecutive edges.  ONE streaming pass over the trace (HBM-bound): 16-byte loads, edge positions
// appended to a list with one atomic per thread that saw an edge (edges are rare; the host sorts the
// few hundred positions), and min/max of every 4096-sample chunk, from which the min/max of the long
// pieces follow without a second pass over their samples.
constexpr int DET_NT = 256;
constexpr int DET_PER = 16;                         // samples per thread
constexpr int DET_CHUNK = DET_NT * DET_PER;

template <int DT>
__device__ __forceinline__ bool below_thr(const DevCfg &c, int k, double thr)
{
    // k is the integer count; pA = k * quantum exactly (fp32 input: k = x / quantum on the grid)
    return static_cast<double>(k) * c.q < thr;
}

template <int DT>
__global__ __launch_bounds__(DET_NT) void edge_scan_kernel(DevCfg c, int64_t n, double thr, int *tics,
                                                           unsigned *n_tics, unsigned tics_cap, int2 *chunk_mm,
                                                           unsigned *status)
{
    __shared__ int smin[DET_NT / 64], smax[DET_NT / 64];
    const int64_t i0 = static_cast<int64_t>(blockIdx.x) * DET_CHUNK + static_cast<int64_t>(threadIdx.x) * DET_PER;
    unsigned bits = 0, bad = 0;
    int mn = 0x7fffffff, mx = static_cast<int>(0x80000000);
    if (i0 < n) {
        int k[DET_PER];
        constexpr int ES = static_cast<int>(sizeof(typename Raw<DT>::type));
        const char *p = static_cast<const char *>(c.samples) + i0 * ES;
        if (i0 + DET_PER <= n && (reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
            constexpr int NV = DET_PER * ES / 16;
            int4 raw[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) raw[v] = reinterpret_cast<const int4 *>(p)[v];
            if (sdt(DT) == PS_DTYPE_F32) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const float f[4] = {__int_as_float(raw[v].x), __int_as_float(raw[v].y), __int_as_float(raw[v].z), __int_as_float(raw[v].w)};
#pragma unroll
                    for (int e = 0; e < 4; ++e) k[v * 4 + e] = to_count<DT>(c, f[e], bad);
                }
            } else {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int w[4] = {raw[v].x, raw[v].y, raw[v].z, raw[v].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        k[v * 8 + e] = ((e & 1) ? (w[e >> 1] >> 16) : static_cast<int>(static_cast<short>(w[e >> 1] & 0xffff))) + c.off_counts;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < DET_PER; ++u) k[u] = (i0 + u < n) ? load_count<DT>(c, i0 + u, bad) : 0;
        }
        bool prev = i0 > 0 ? below_thr<DT>(c, load_count<DT>(c, i0 - 1, bad), thr) : false;
#pragma unroll
        for (int u = 0; u < DET_PER; ++u) {
            if (i0 + u < n) {
                const bool m = below_thr<DT>(c, k[u], thr);
                if (i0 + u > 0 && m != prev) bits |= 1u << u;
                prev = m;
                mn = min(mn, k[u]); mx = max(mx, k[u]);
            }
        }
    }
    if (bits) {
        const unsigned cnt = __popc(bits);
        unsigned o = atomicAdd(n_tics, cnt);
        for (int u = 0; u < DET_PER; ++u)
            if (bits & (1u << u)) { if (o < tics_cap) tics[o] = static_cast<int>(i0 + u); ++o; }
    }
    wave_minmax(mn, mx);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 63) { smin[wave] = mn; smax[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < DET_NT / 64; ++w) { mn = min(mn, smin[w]); mx = max(mx, smax[w]); }
        chunk_mm[blockIdx.x] = make_int2(min(mn, smin[0]), max(mx, smax[0]));
    }
    if (bad) atomicOr(status, bad);
}

// min / max (counts) of every candidate piece [a, b): whole chunks from the table, the two ragged ends
// sample by sample.  One workgroup per piece.
template <int DT>
__global__ __launch_bounds__(256) void piece_minmax_kernel(DevCfg c, const int2 *cand, int n_cand, const int2 *chunk_mm,
                                                           int2 *mm)
{
    __shared__ int smin[4], smax[4];
    if (static_cast<int>(blockIdx.x) >= n_cand) return;
    const int2 pc = cand[blockIdx.x];
    int mn = 0x7fffffff, mx = static_cast<int>(0x80000000);
    unsigned bad = 0;
    const int64_t c0 = (static_cast<int64_t>(pc.x) + DET_CHUNK - 1) / DET_CHUNK, c1 = pc.y / DET_CHUNK;   // full chunks [c0, c1)
    if (c0 < c1) {
        for (int64_t b = c0 + threadIdx.x; b < c1; b += 256) { const int2 v = chunk_mm[b]; mn = min(mn, v.x); mx = max(mx, v.y); }
        for (int64_t i = pc.x + threadIdx.x; i < c0 * DET_CHUNK; i += 256) { const int k = load_count<DT>(c, i, bad); mn = min(mn, k); mx = max(mx, k); }
        for (int64_t i = c1 * DET_CHUNK + threadIdx.x; i < pc.y; i += 256) { const int k = load_count<DT>(c, i, bad); mn = min(mn, k); mx = max(mx, k); }
    } else {
        for (int64_t i = pc.x + threadIdx.x; i < pc.y; i += 256) { const int k = load_count<DT>(c, i, bad); mn = min(mn, k); mx = max(mx, k); }
    }
    wave_minmax(mn, mx);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 63) { smin[wave] = mn; smax[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = smin[0], b = smax[0];
        for (int w = 1; w < 4; ++w) { a = min(a, smin[w]); b = max(b, smax[w]); }
        mm[blockIdx.x] = make_int2(a, b);
    }
}

// The detector of the single-pass file route (round 6): K0 ran over the whole trace as one event and judged every 8-sample block
// against the threshold (DevCfg::blk_cls: 0 all at or above, 1 all below, 2 mixed -- 2 bits per block, 3 MB per 1e8 samples) and
// left min / max per 128 blocks (DevCfg::cls_mm).  A thread takes 64 blocks (16 bytes of classes): uniform and equal to the
// block before them -- nearly always -- there is nothing to do.  Else: a block whose samples lie on one side is judged by its
// class; a mixed one (the slopes of the events, a handful per 1e8 samples) by its 8 samples, with the detector's own predicate
// (below_thr).  Position i (an edge sits at i when below(i) != below(i-1), parsers.py:142-147) is judged exactly once: by its own
// block when that is mixed, by the mixed block before it when its own is not, by the class comparison when neither is.  Output as
// edge_scan_kernel's: the edge list and min / max (counts) per DET_CHUNK samples.
constexpr int CLS_NT = 256;
template <int DT>
__global__ __launch_bounds__(CLS_NT) void edge_cls_kernel(DevCfg c, int64_t n, double thr, const unsigned char *cls, const int2 *cls_mm,
                                                          const int4 *ev_info0, int *tics, unsigned *n_tics, unsigned tics_cap,
                                                          int2 *chunk_mm, unsigned *status)
{
    const long long nb = (n + 7) >> 3;
    const long long tid = static_cast<long long>(blockIdx.x) * CLS_NT + threadIdx.x;
    const int m = ev_info0[0].x;                        // the trace's centre: K0 wrote it
    unsigned bad = 0;
    auto emit = [&](long long i) {
        const unsigned o = atomicAdd(n_tics, 1u);
        if (o < tics_cap) tics[o] = static_cast<int>(i);
    };
    const long long b0 = 64 * tid;
    uint4 w4 = make_uint4(0u, 0u, 0u, 0u);
    int prev_cls = -1;
    bool quiet = true;
    if (b0 < nb) {
        w4 = reinterpret_cast<const uint4 *>(cls)[tid];
        prev_cls = b0 > 0 ? (cls[(b0 >> 2) - 1] >> 6) & 3 : -1;
        const unsigned pat = prev_cls == 1 ? 0x55555555u : 0u;
        quiet = prev_cls >= 0 && prev_cls != 2 && b0 + 64 <= nb && w4.x == pat && w4.y == pat && w4.z == pat && w4.w == pat;
    }
#ifdef PS_DIAG
    if (c.rep_stage == 7) quiet = true;                 // (experiments: what the kernel costs without its slow path)
#endif
    // The lanes whose 64 blocks hold an edge (a few hundred per 1e8 samples) are served one after the other BY THE WHOLE WAVE, a
    // block per lane: left to itself such a lane would walk its 64 blocks alone -- ~4 000 instructions of one lane in a wave
    // that issues one every few cycles, 26 of the kernel's 31 us (measured, docs/ROUND_6.md).
    const int lane = threadIdx.x & 63;
    unsigned long long need = __ballot(!quiet);
    while (need) {
        const int src = __ffsll(static_cast<long long>(need)) - 1;
        need &= need - 1;
        const unsigned x0 = __shfl(w4.x, src), x1 = __shfl(w4.y, src), x2 = __shfl(w4.z, src), x3 = __shfl(w4.w, src);
        const int sprev = __shfl(prev_cls, src);
        const long long sb0 = 64 * (tid - lane + src);
        const long long b = sb0 + lane;
        if (b < nb) {
            auto cls_j = [&](int j) -> int {                // class of block sb0 + j, 0 <= j < 64
                const int q = j >> 4;
                const unsigned wq = q == 0 ? x0 : q == 1 ? x1 : q == 2 ? x2 : x3;
                return (wq >> (2 * (j & 15))) & 3;
            };
            const int cb = cls_j(lane);
            const int cp = lane > 0 ? cls_j(lane - 1) : sprev;               // (-1: there is no block before the trace's first)
            if (cb != 2) {
                if (cp >= 0 && cp != 2 && cp != cb) emit(8 * b);
            } else {
                const long long i0 = 8 * b, i1 = min(i0 + 8, static_cast<long long>(n));
                int k[9];                                   // the block's samples and the one before them: loaded together, judged afterwards
    #pragma unroll
                for (int u = 0; u < 9; ++u) k[u] = (i0 + u - 1 >= 0 && i0 + u - 1 < i1) ? load_count<DT>(c, i0 + u - 1, bad) : 0;
                bool prev = i0 > 0 ? below_thr<DT>(c, k[0], thr) : false;
    #pragma unroll
                for (int u = 1; u < 9; ++u) {
                    const long long i = i0 + u - 1;
                    if (i < i1) {
                        const bool mi = below_thr<DT>(c, k[u], thr);
                        if (i > 0 && mi != prev) emit(i);
                        prev = mi;
                    }
                }
                if (b + 1 < nb) {
                    const int cn = lane < 63 ? cls_j(lane + 1) : (cls[(b + 1) >> 2] >> (2 * ((b + 1) & 3))) & 3;
                    if (cn != 2 && (cn == 1) != prev) emit(i0 + 8);
                }
            }
        }
    }
    // min / max per DET_CHUNK samples from K0's per 128 blocks (blocks beyond the trace were left out there)
    constexpr int KPC = DET_CHUNK / (8 * BS_CHUNK);     // K0 chunks per detector chunk
    const long long n_det = (n + DET_CHUNK - 1) / DET_CHUNK;
    if (tid < n_det) {
        int a = 0x7fffffff, z = static_cast<int>(0x80000000);
        for (int k = 0; k < KPC; ++k) {
            const long long kc = tid * KPC + k;
            if (kc * BS_CHUNK < nb) { const int2 mm = cls_mm[kc]; a = min(a, mm.x); z = max(z, mm.y); }
        }
        chunk_mm[tid] = make_int2(a + m, z + m);
    }
    if (bad) atomicOr(status, bad);
}

// start of the single-pass call: the status block cleared and the one-event table of the whole trace, in one launch
__global__ __launch_bounds__(256) void trace_setup_kernel(unsigned long long *small, int n_words, long long *ev, long long n, long long nb)
{
    for (int i = threadIdx.x; i < n_words; i += 256) small[i] = 0ull;
    if (threadIdx.x == 0) { ev[0] = 0; ev[1] = n; ev[2] = 0; ev[3] = nb; }    // ev_off[0], ev_len[0], ev_boff[0..1]
}

// the events cut out of a trace whose digest is trace-aligned: (centre of the trace, phase, first block) per event
__global__ __launch_bounds__(256) void ev_info_trace_kernel(const int64_t *ev_start, int n_ev, const int4 *ev_info0, int4 *ev_info)
{
    const int m = ev_info0[0].x;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n_ev; e += gridDim.x * 256) {
        const long long s0 = ev_start[e], b = s0 >> 3;
        ev_info[e] = make_int4(m, static_cast<int>(s0 & 7), static_cast<int>(b & 0xffffffffLL), static_cast<int>(b >> 32));
    }
}

// ---- K2: per-segment statistics, core.py:209-223 ------------------------------------------------
// One workgroup per segment.  Sums of counts and counts^2 are exact (fp64 holds the integers),
// mean = q*S1/n, std = q*sqrt(S2/n - (S1/n)^2) (population), min/max exact.
constexpr int STAT_NT = 256;
template <int DT>
__global__ __launch_bounds__(STAT_NT) void segstat_kernel(DevCfg c, const int64_t *ev_start, const int64_t *ev_len, int32_t n_ev,
                                                          const int32_t *bounds, const int64_t *bounds_off,
                                                          ps_segstat *stats, unsigned *status)
{
    constexpr int NW = STAT_NT / 64;
    __shared__ double ws1[NW], ws2[NW];
    __shared__ int smin[NW], smax[NW];
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
    const int n_e = static_cast<int>(ev_len[e]);
    const int a = s == 0 ? 0 : bounds[boff + s - 1];
    const int b = s == cnt ? n_e : bounds[boff + s];
    const int64_t g0 = ev_start[e];
    unsigned bad = 0;
    double s1 = 0, s2 = 0;
    int mn = 0x7fffffff, mx = static_cast<int>(0x80000000);
    for (int i = a + threadIdx.x; i < b; i += STAT_NT) {
        int k = load_count<DT>(c, g0 + i, bad);
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
    if (lane == 0) { ws1[wave] = s1; ws2[wave] = s2; smin[wave] = mn; smax[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t1 = 0, t2 = 0;
        int tm = 0x7fffffff, tx = static_cast<int>(0x80000000);
        for (int w = 0; w < NW; ++w) {
            t1 += ws1[w]; t2 += ws2[w];
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

// ---- diagnostic: the bounds of the scan against the gains they cover (ps_audit_bounds; tests only) ---------------------
// One wave per window [win[i].x, win[i].y) of event 0, candidates as the recursion would give them (min_width from both ends).
template <int DT>
__global__ __launch_bounds__(64) PS_SCAN_REGS void audit_kernel(DevCfg c, const int2 *win, int n_win, unsigned *status)
{
    __shared__ SharedT<64> sh;
    Work wk = PS_WORK_INIT;
    unsigned bad = 0;
    const EvRef er = ev_ref_of(c, 0);
    for (int i = blockIdx.x; i < n_win; i += gridDim.x) {
        const int ps = uni(win[i].x), pe = uni(win[i].y);
        (void)scan_window_bs<DT, true, true>(c, er, 0, ps, pe, ps + c.mw, pe - c.mw, c.min_gain, sh, bad, wk);
    }
    if (bad) atomicOr(status, bad);
}

}  // namespace ps

#include "seg_filter.hpp"
