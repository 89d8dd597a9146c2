// poreseg.hip -- C ABI (include/poreseg.h) and host orchestration of libporeseg.so.
//
// Pipeline of ps_segment_batch, one stream, one host sync at the end:
//   phase 0  blocksum_kernel   K0: chunk-prefixed sums per 8-sample block (the only pass over the samples)
//   phase 1  spine_kernel      one workgroup per tile: speculative spine of rec(T, END)
//   phase 1b bridge_kernel     continues each tile's chain until it meets the downstream tile's
//   stitch   assemble_* kernels  true spine per event from the tile spines (host-side only as the fallback)
//   phase 3  tree_kernel       one job per true spine step: rec(a_k, a_{k+1}) in order
//   gather   item_scan + gather kernels -> contiguous, sorted boundary list per event
//   K2       segstat_kernel (optional)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <complex>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "poreseg.h"
#include "seg_device.hpp"
#include "seg_align.hpp"

using namespace ps;

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool owned = true;        // false: p points into another buffer (alias), nothing to free
    hipError_t reserve(size_t bytes)
    {
        if (!owned) { p = nullptr; cap = 0; owned = true; }
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
        size_t want = std::max(bytes, cap + cap / 2);
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    // as reserve(), but a fresh allocation is cleared (buffers whose stale contents are compared with an epoch)
    hipError_t reserve_zeroed(size_t bytes, hipStream_t st)
    {
        const void *before = p;
        const size_t cap0 = cap;
        hipError_t e = reserve(bytes);
        if (e == hipSuccess && (p != before || cap != cap0) && p) e = hipMemsetAsync(p, 0, cap, st);
        return e;
    }
    // Points this buffer into `blob` (one upload for several small arrays); a later reserve() allocates afresh.
    void alias(void *ptr) { if (p && owned) (void)hipFree(p); p = ptr; cap = 0; owned = false; }
    void release() { if (p && owned) (void)hipFree(p); p = nullptr; cap = 0; owned = true; }
    template <class T> T *as() { return static_cast<T *>(p); }
};

struct HostBuf {          // pinned staging memory
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        size_t want = std::max(bytes, cap + cap / 2);
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() { return static_cast<T *>(p); }
};

}  // namespace

struct ps_ctx {
    int device = 0;
    int n_cu = 0;             // compute units (queried once)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    int64_t tile_len = 0, halo = 0;
    int mode = MODE_FAST, spine_nt = 512, tree_nt = 256;
    int lds_max_samples = 0;
    int rep_eval = 1, rep_stage = 1, rep_sum = 1;
    uint8_t *d_is_spine = nullptr;   // optional output of the current call
    int prune = 1;
    double wide_quantum = 0;  // quantum of the last call K0 refused (counts too wide) ...
    int wide_skip = 0;        // ... and the number of calls with that quantum that still start where that call ended:
    int wide_mode = 0;        // ... 2 = block-sum scan on the 64-bit digest, 0 = LDS-window scan
    int tree_mw = 0;          // 1: block-sum tree kernel with TREE_W waves per workgroup sharing their job list (round 3: single-wave workgroups are faster at four waves per SIMD and need no spills)
    int upload_by_kernel = 1; // 1: the call's host tables are fetched by a kernel (no SDMA hand-over), 0: hipMemcpyAsync
    int filter_fused = 1;     // 1: fast filters run both directions in one kernel over tiles with halos, 0: always the exact three-pass scan
    int k0_waves = 0;         // > 0: K0 runs as a persistent kernel, this many waves per SIMD (twice that for int16 samples) striding over the call with
                              // their next wave block's samples in flight (round 5) -- what a context that shares the chip wants (engine.StreamPool
                              // sets 1: 0.193 -> 0.170 ms per step with sixteen calls in flight); 0: one wave per wave block, as many as the registers
                              // allow -- faster for a call that has the chip to itself (1e9 samples: K0 0.91 against 0.98 ms)
    float noise_k = 0.1f;     // near-tie accounting of the wide route (DevCfg::noise_k; PORESEG_NOISE_K)
    int k0_admit = 0;         // > 0: at most this many calls of the device have their K0 in flight (K0Gate); 0: no limit
    int k0_sets = 2;          // register sets of a persistent K0 wave (seg_bs.hpp: blocksum_kernel<DT, NS>); 3 / 4 in the diagnostic library only (rejected)
    bool chain_held = false;  // this call has a ticket of the device's K0 chain (K0Chain) and has not recorded its event yet
    unsigned long long chain_ticket = 0;
    bool defer_sync = false;  // (internal) ps_filter_requantise_batch: the filter entry only queues its kernels -- no status clear, copy, sync
    int scan_lds_pad = 0;     // diagnostics (libporeseg_diag.so only: option scan_lds_pad): unused dynamic LDS per single-wave scan workgroup -- caps the scan waves per SIMD
    int k0_unaligned = 1;     // K0's fast route loads 16 bytes from sample-aligned addresses: probed once per device at ps_create (k0_unaligned_probe);
                              // option k0_unaligned 0 restores the 16-byte condition of rounds 1-4 (tests)
    int single_pass = 1;      // 1 (round 6): ps_detect_segment_trace runs K0 once over the whole trace (detector + every event from one digest), 0: the two calls
    int gather_fused = 1;     // 1 (round 6): the gather places its items from per-256-job count sums (gather_scan_kernel), no item_scan_kernel; 0: rounds 2-5
    int download_by_kernel = 1;   // 1 (round 6): status block + per-event offsets go back by a kernel writing pinned memory, 0: hipMemcpyAsync
    int debug = 0;            // option debug: the library says on stderr which seams gave up, which occupancy it found (prints only; results unchanged)
#ifdef PS_DIAG
    // Diagnostics that return WRONG or stale results exist in libporeseg_diag.so only (make -C pypore_amd/csrc diag): the product
    // library neither holds this code nor reads the environment (round 6).
    int dbg_phase = 0;        // 1 a call that repeats the previous one's layout skips K0 (the digest is still there: what the scan kernels cost
                              // on their own), 2 K0 only
    int dbg_k0_nogrp = 0;     // 1 a call that repeats the previous one's layout runs K0 WITHOUT its group-record part (the records of the previous
                              // call are still there and the scans read those: what K0's ~37 instructions per block cost the step)
#endif
    int k0_shared = 0;        // 1: upload + K0 of this context run on the device's shared FRONT stream (round 5): the K0 launches of all contexts that
                              // ask for it are serialised there, back to back, and the context's own stream takes over behind an event.  K0 is the
                              // one kernel of a call that is bound by HBM; several of them at once only share the same bytes per second while the
                              // scan kernels behind each of them wait -- in a row, call i's scans run under call i+1's K0.  engine.StreamPool sets it.
    hipEvent_t ev_front[2] = {};   // hand-over events (no timing): [0] context stream -> front stream, [1] front stream -> context stream
    bool stream_idle = true;  // the context's stream had nothing queued when the current call began (no hand-over to the front stream needed)
    int tree_par = 1;         // 1: subtree jobs on the 64-bit digest (filtered events: deep recursions) are shared by the waves of a workgroup (tree_par_kernel), 0: one wave per job
    int groups = 1;           // 1: K0 writes group records and the window scans start with the coarse pass over them (narrow digest), 0: every row is swept
    int scan_bs = 1;          // 1: block-sum scan with single-wave workgroups (seg_bs.hpp), 0: LDS-window scan
    int wide_bs = 1;          // 1: counts too wide for the 32-bit digest are retried on the 64-bit digest, 0: straight to the LDS-window scan
    int bridge_single = 1 << 30;   // anchors a single-wave bridge adds before it hands the seam to the look-ahead kernel (measured: handing over early is slower)
    int tree_tail_pct = 0;    // tree_mw_kernel: share of the job list drawn dynamically (counter in HBM) at the end
    // Share of the resident wave slots the single-wave scan kernels are launched on (experiments; the subtree kernel
    // decides on the device how many of its slots work: tree_jobs_per_wave, seg_device.hpp: tree_kernel).  Measured on
    // the bench trace with 100 / 75 / 62 / 50 / 44 / 37 %: four calls in flight 0.274 / 0.253 / 0.250 / 0.2445 / 0.257 /
    // 0.270 ms per step, one call 0.522 / 0.499 / 0.511 / 0.505 / 0.572 / 0.563 ms (below 50 % the spine kernel's 2 048
    // tiles are no longer all resident); on the 1e9-sample trace 50 % costs 23 % (spine 0.76 -> 1.12 ms, subtrees 0.68 ->
    // 0.90): with many jobs per slot four waves per SIMD deliver 1.4 x the throughput of two.
    int slots_pct = 100;
    int tree_jobs_per_wave = 4;   // subtree kernel: jobs / this many slots work, between half and all of them (0: all)
    // host-side caches: occupancy per kernel, dynamic-LDS attribute last set, the tile tables of the last call
    struct OccKey { const void *fn; int nt; size_t lds; unsigned slots; };
    std::vector<OccKey> occ_cache;
    std::vector<std::pair<const void *, int>> attr_cache;
    struct TileCache {
        bool valid = false;
        int32_t n_ev = 0; int mw = 0, W = 0; int64_t L = 0; bool use_bs = false;
        std::vector<int64_t> ev_start, ev_len;
        size_t nj = 0, jb = 0, up_bytes = 0; int64_t list_entries = 0, total_len = 0, sample_end = 0, nb_total = 0;
    } tile_cache;
    DevBuf bsum, ev_info, chunk_mabs, ev_boff, blk_mm, grp, filt_fwd, filt_agg, filt_zin, up_dev;
    DevBuf align_in, align_scratch;
    DevBuf ev_info_tr;        // single-pass file route (ps_detect_segment_trace): (centre, phase, first block) of the events cut out of a trace-aligned digest
    DevBuf blk_cls, cls_mm;   // ... K0's verdict per block against the detector's threshold (2 bits), min / max per 128 blocks
    DevBuf pre_c;             // exact route (ps_segment_exact_f64): c and c2 of the call's samples, 16 B per sample
    int stitch_host = 0;      // 1: host stitch with halo tiles (the fallback path) always
    DevBuf ev_len, det_counts, det_tics, det_cand;
    DevBuf bridges, bmeta, tile_i32, sp_off, spine_items, asm_hdr, ev_first_tile;
    DevBuf bridge_ext, ext_slot, ext_list;          // second chance for seams that gave up (seg_device.hpp: EXT_MAX)
    DevBuf lat_state, lat_seam, lat_res;            // helpers of the look-ahead kernel (seg_device.hpp: LAT_D)
    unsigned lat_tag_next = 1;                      // tags handed to calls so far (24 bits; the results are cleared before they run out)
    unsigned lat_tag_base = 0;                      // this call's
    int lat_help = 1;                               // option lat_help / PORESEG_LAT_HELP: 0 = every seam walks alone, as before round 5
    int bridge_budget = BR_MAX;                     // option bridge_budget (1 .. BR_MAX): anchors before a seam gives up -- tests lower it to reach the second chance on small inputs
    int bridge_ext_on = 1;                          // option bridge_ext / PORESEG_BRIDGE_EXT: 0 = straight to the host stitch, as before round 5
    HostBuf h_hdr;
    DevBuf spine_jobs, spine_scratch, spine_dense, spine_meta, tree_jobs, tree_scratch, tree_spill,
        tree_counts, items, item_pos, first_item, ev_off, bounds_off, small;
    HostBuf h_meta, h_dense, h_small, h_up;
    hipEvent_t ev[10] = {};   // [0..7] phase marks (timing level 2), [8] start and [9] end of the call's device work (level 1)
    int timing = 1;           // 0: no events, 1: start/end of the sequence, 2: an event between the phases as well (each costs
                              // ~6 us of idle GPU: the next kernel does not start back to back)
    double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t counters[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // [12] chunk results the look-ahead helpers published, [13] of which an owner took
};

namespace {

int fail(ps_ctx *ctx, int code, const char *fmt, ...)
{
    if (ctx) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        ctx->err = buf;
    }
    return code;
}

#define HIP_TRY(ctx, expr)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(ctx, PS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// The front stream of a device (ps_ctx::k0_shared): one stream per device and process on which the upload + K0 launches of
// all contexts are queued back to back (the mutex keeps one call's pair of launches and its event together).
struct FrontStream { std::mutex mu; hipStream_t s = nullptr; };
FrontStream g_front[16];
FrontStream *front_for(ps_ctx *ctx)
{
    if (ctx->device < 0 || ctx->device >= 16) return nullptr;
    FrontStream *f = &g_front[ctx->device];
    std::lock_guard<std::mutex> lk(f->mu);
    if (!f->s) {
#ifdef PS_DIAG
        int lo = 0, hi = 0;
        const char *pr = std::getenv("PORESEG_FRONT_PRIORITY");         // 1: highest priority the device offers (experiments)
        if (pr && std::atoi(pr) != 0 && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess) {
            if (hipStreamCreateWithPriority(&f->s, hipStreamNonBlocking, hi) != hipSuccess) { f->s = nullptr; (void)hipGetLastError(); }
        }
#endif
        if (!f->s && hipStreamCreateWithFlags(&f->s, hipStreamNonBlocking) != hipSuccess) { f->s = nullptr; (void)hipGetLastError(); return nullptr; }
    }
    return f;
}

// K0 admission (ps_ctx::k0_admit): at most M calls of a device have their K0 in flight at a time.  K0 is bound by HBM: T of them
// side by side only share the same bytes per second, and all of them -- with every call's scan kernels behind them -- finish late
// together (sixteen contexts: 0.188 -> 0.175 ms per step with M = 3, round 5).  Round 6: ON THE DEVICE.  The calls of a device
// take tickets; call t queues, in front of its K0, a wait for the event recorded behind the K0 of call t - M (hipStreamWaitEvent:
// a barrier packet in its own queue).  Round 5's gate had host threads poll hipEventQuery every 25 us under one mutex and could
// spin forever on a faulted device; here the host thread never waits, so there is nothing to time out or to leak: a
// call that fails between its ticket and its record leaves a ticket nobody finds recorded, and whoever looks for it does not
// wait.  Same step time as the polling gate (0.1709 against 0.1690 ms, 5 interleaved pairs: profiles/r06_experiments).
struct K0Chain {
    static constexpr int RING = 64;
    std::mutex mu;
    hipEvent_t ev[RING] = {};
    unsigned long long gen[RING];                       // ticket whose K0 the event was last recorded behind (+1; 0: never)
    unsigned long long next = 0;
    bool made = false;
};
K0Chain g_chain[16];
int chain_enter(ps_ctx *ctx, int device, int max_in_flight, hipStream_t st, unsigned long long *ticket)
{
    K0Chain &c = g_chain[device & 15];
    std::lock_guard<std::mutex> lk(c.mu);
    if (!c.made) {
        for (int i = 0; i < K0Chain::RING; ++i) { HIP_TRY(ctx, hipEventCreateWithFlags(&c.ev[i], hipEventDisableTiming)); c.gen[i] = 0; }
        c.made = true;
    }
    const unsigned long long t = c.next++;
    *ticket = t;
    const int m = std::min(max_in_flight, K0Chain::RING / 2);
    if (t >= static_cast<unsigned long long>(m)) {
        const unsigned long long w = t - static_cast<unsigned long long>(m);
        if (c.gen[w % K0Chain::RING] == w + 1) HIP_TRY(ctx, hipStreamWaitEvent(st, c.ev[w % K0Chain::RING], 0));
    }
    return PS_OK;
}
int chain_publish(ps_ctx *ctx, int device, hipStream_t st, unsigned long long ticket)     // K0 has been queued on st
{
    K0Chain &c = g_chain[device & 15];
    std::lock_guard<std::mutex> lk(c.mu);
    HIP_TRY(ctx, hipEventRecord(c.ev[ticket % K0Chain::RING], st));
    c.gen[ticket % K0Chain::RING] = ticket + 1;
    return PS_OK;
}
// status word + work counters live in ctx->small: [0] status (u32, padded to 8), [1..3] work, [4] dense count
// (hdr: the stitch header lives behind the counters so that one copy brings both back)
// (SMALL_TAIL bytes behind it hold the per-event offsets of small batches, so that one copy brings everything back)
constexpr size_t SMALL_TAIL = 64 * 1024;
struct SmallLayout { unsigned long long status, work0, work1, work2, dense, stamp[12], life[9], qctl, qhead, tree_tail; AsmHeader hdr;
                     unsigned long long lat_ctl[6]; int lat_prog[LAT_D]; };     // (helpers of the look-ahead kernel, seg_device.hpp: LAT_D)

int make_cfg(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt, int mw, int maxw, int W,
             double min_gain, DevCfg *c)
{
    if (!fmt) return fail(ctx, PS_ERR_ARG, "sample format is NULL");
    if (fmt->dtype != PS_DTYPE_F32 && fmt->dtype != PS_DTYPE_I16)
        return fail(ctx, PS_ERR_ARG, "unknown dtype %d", fmt->dtype);
    if (!(fmt->quantum > 0) || !std::isfinite(fmt->quantum))
        return fail(ctx, PS_ERR_ARG, "quantum must be positive and finite");
    c->samples = d_samples;
    c->dtype = fmt->dtype;
    // (fp32 samples are absolute values: offset_counts is not added to them -- it names the level the caller subtracted
    //  upstream, include/poreseg.h)
    c->off_counts = fmt->dtype == PS_DTYPE_F32 ? 0 : fmt->offset_counts;
    c->dc_counts = fmt->dtype == PS_DTYPE_F32 ? static_cast<double>(fmt->offset_counts) : 0.0;
    c->noise_k = ctx->noise_k;
    c->inv_q = static_cast<float>(1.0 / fmt->quantum);
    c->q = fmt->quantum;
    c->q2 = fmt->quantum * fmt->quantum;
    c->mw = mw; c->maxw = maxw; c->W = W; c->half = W / 2;
    c->min_gain = min_gain;
    c->mode = ctx->mode;
    c->prune = ctx->prune;
    c->lds_cap = std::max(1, std::min(W, ctx->lds_max_samples));
    c->bsum = nullptr; c->ev_info = nullptr; c->chunk_tot = nullptr; c->blk_mm = nullptr; c->grp = nullptr; c->bs_wide = 0;
    c->blk_cls = nullptr; c->cls_mm = nullptr; c->cls_kthr = 0;
    c->k0_unaligned = ctx->k0_unaligned;
    c->pre_c = nullptr; c->pre_c2 = nullptr;
    c->dbg = ctx->small.as<SmallLayout>()->stamp;
    c->rep_eval = ctx->rep_eval; c->rep_stage = ctx->rep_stage; c->rep_sum = ctx->rep_sum;
    return PS_OK;
}

// dynamic LDS of the scan kernels: window samples (+ alignment slack) and the per-block sums of
// the pruned screen (one int2 per PBLK samples, at most one extra slot per thread)
inline size_t lds_bytes_for(int lds_cap, int nt)
{
    return (((static_cast<size_t>(lds_cap) + 32) * sizeof(lds_t) + 15) & ~static_cast<size_t>(15)) +
           (static_cast<size_t>(lds_cap) / PBLK + static_cast<size_t>(nt) + 8) * sizeof(int2);
}

// LDS budget: 160 KB per CU minus the static Shared block and a little slack
constexpr int LDS_BYTES_MAX = 160 * 1024 - static_cast<int>(sizeof(Shared)) - 512 - 64;

// Resident workgroups of a kernel on this device (occupancy x CUs): the scan kernels are launched with one
// workgroup per slot and stride over their jobs.
template <typename K> unsigned resident_slots(ps_ctx *ctx, K kernel, int nt, size_t lds)
{
    const void *fn = reinterpret_cast<const void *>(kernel);
    for (const auto &k : ctx->occ_cache)
        if (k.fn == fn && k.nt == nt && k.lds == lds) return k.slots;
    int per_cu = 0;
    if (ctx->n_cu <= 0) {
        hipDeviceProp_t prop;
        ctx->n_cu = hipGetDeviceProperties(&prop, ctx->device) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, nt, lds) != hipSuccess || per_cu <= 0)
        per_cu = 1;
    unsigned slots = static_cast<unsigned>(per_cu) * static_cast<unsigned>(ctx->n_cu);
    if (ctx->debug) fprintf(stderr, "[poreseg] occupancy: %d workgroups of %d threads per CU (dynamic LDS %zu), %d CUs\n", per_cu, nt, lds, ctx->n_cu);
    if (nt <= 256) slots = std::max(1u, static_cast<unsigned>(static_cast<unsigned long long>(slots) * static_cast<unsigned>(ctx->slots_pct) / 100u));
    ctx->occ_cache.push_back({fn, nt, lds, slots});
    return slots;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) only when the value changes (it is a driver call on every launch otherwise)
hipError_t set_dyn_lds(ps_ctx *ctx, const void *fn, int lds)
{
    for (auto &a : ctx->attr_cache)
        if (a.first == fn) {
            if (a.second == lds) return hipSuccess;
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e == hipSuccess) a.second = lds;
            return e;
        }
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) ctx->attr_cache.push_back({fn, lds});
    return e;
}

template <int NT, int DT> int launch_spine(ps_ctx *ctx, const DevCfg &cfg, unsigned nj, SmallLayout *sm, bool list_mode = false)
{
    const size_t lds = NT == 64 ? static_cast<size_t>(ctx->scan_lds_pad) : lds_bytes_for(cfg.lds_cap, NT);
    HIP_TRY(ctx, set_dyn_lds(ctx, reinterpret_cast<const void *>(spine_kernel<NT, DT>), static_cast<int>(lds)));
    const unsigned grid = std::min(nj, resident_slots(ctx, spine_kernel<NT, DT>, NT, lds));
    hipLaunchKernelGGL((spine_kernel<NT, DT>), dim3(grid), dim3(NT), lds, ctx->stream, cfg,
                       ctx->spine_jobs.as<SpineJob>(), ctx->spine_scratch.as<int2>(),
                       list_mode ? nullptr : ctx->spine_dense.as<int2>(), ctx->spine_meta.as<int4>(), &sm->dense, reinterpret_cast<unsigned *>(&sm->status), &sm->work0,
                       static_cast<int>(nj));
    HIP_TRY(ctx, hipGetLastError());
    return PS_OK;
}

template <int NT, int DT> int launch_tree(ps_ctx *ctx, const DevCfg &cfg, unsigned nj, SmallLayout *sm, size_t n_jobs,
                                          const AsmHeader *d_hdr, int par_max_jobs = 0)
{
    const size_t lds = NT == 64 ? static_cast<size_t>(ctx->scan_lds_pad) : lds_bytes_for(cfg.lds_cap, NT);
    HIP_TRY(ctx, set_dyn_lds(ctx, reinterpret_cast<const void *>(tree_kernel<NT, DT>), static_cast<int>(lds)));
    const unsigned grid = std::min(nj, resident_slots(ctx, tree_kernel<NT, DT>, NT, lds));
    hipLaunchKernelGGL((tree_kernel<NT, DT>), dim3(grid), dim3(NT), lds, ctx->stream, cfg,
                       ctx->tree_jobs.as<TreeJob>(), ctx->tree_scratch.as<int32_t>(), ctx->tree_spill.as<int2>(),
                       ctx->tree_counts.as<int32_t>(), reinterpret_cast<unsigned *>(&sm->status), &sm->work0,
                       static_cast<long long>(n_jobs), d_hdr,
                       // (not on the 64-bit digest: those are filtered events, whose jobs hold many windows each -- 32 of them:
                       //  subtree kernel 1.30 ms on all slots, 1.71 ms with the rule)
                       NT == 64 && !(DT & DT_WIDE) ? ctx->tree_jobs_per_wave : 0, par_max_jobs);
    HIP_TRY(ctx, hipGetLastError());
    return PS_OK;
}

// block-sum scan, TREE_W waves per workgroup sharing the workgroup's job list (tree_mw_kernel)
template <int DT> int launch_tree_mw(ps_ctx *ctx, const DevCfg &cfg, unsigned nj, SmallLayout *sm, size_t n_jobs,
                                     const AsmHeader *d_hdr)
{
    const unsigned want = (nj + TREE_W - 1) / TREE_W;
    const size_t lds = sizeof(SharedT<64>) * TREE_W;
    HIP_TRY(ctx, set_dyn_lds(ctx, reinterpret_cast<const void *>(tree_mw_kernel<DT>), static_cast<int>(lds)));
    const unsigned grid = std::max(1u, std::min(want, resident_slots(ctx, tree_mw_kernel<DT>, 64 * TREE_W, lds)));
    hipLaunchKernelGGL((tree_mw_kernel<DT>), dim3(grid), dim3(64 * TREE_W), lds, ctx->stream, cfg,
                       ctx->tree_jobs.as<TreeJob>(), ctx->tree_scratch.as<int32_t>(), ctx->tree_spill.as<int2>(),
                       ctx->tree_counts.as<int32_t>(), reinterpret_cast<unsigned *>(&sm->status), &sm->work0,
                       static_cast<long long>(n_jobs), d_hdr, &sm->tree_tail, ctx->tree_tail_pct);
    HIP_TRY(ctx, hipGetLastError());
    return PS_OK;
}

// deep jobs (filtered events on the 64-bit digest): the PAR_W waves of a workgroup share one job (tree_par_kernel)
template <int DT> int launch_tree_par(ps_ctx *ctx, const DevCfg &cfg, unsigned nj, SmallLayout *sm, size_t n_jobs,
                                      const AsmHeader *d_hdr)
{
    const size_t lds = sizeof(SharedT<64>) * PAR_W + sizeof(ParQ);
    HIP_TRY(ctx, set_dyn_lds(ctx, reinterpret_cast<const void *>(tree_par_kernel<DT>), static_cast<int>(lds)));
    const unsigned slots = resident_slots(ctx, tree_par_kernel<DT>, 64 * PAR_W, lds);
    const unsigned grid = std::max(1u, std::min(nj, slots));
    // calls with at most one job per workgroup slot are this kernel's; the others go to tree_kernel (decided on the device,
    // where the job count is known: both are launched, one of them returns at once)
    const int par_max = static_cast<int>(slots);
    hipLaunchKernelGGL((tree_par_kernel<DT>), dim3(grid), dim3(64 * PAR_W), lds, ctx->stream, cfg,
                       ctx->tree_jobs.as<TreeJob>(), ctx->tree_scratch.as<int32_t>(), ctx->tree_spill.as<int2>(),
                       ctx->tree_counts.as<int32_t>(), reinterpret_cast<unsigned *>(&sm->status), &sm->work0,
                       static_cast<long long>(n_jobs), d_hdr, par_max);
    HIP_TRY(ctx, hipGetLastError());
    return launch_tree<64, DT>(ctx, cfg, nj, sm, n_jobs, d_hdr, par_max);
}

struct Anchor { int32_t pos, kind; };

struct TileList {
    int32_t start = 0;
    std::vector<Anchor> a;
    bool ended = false;
};

int check_status(ps_ctx *ctx, unsigned st)
{
    if (st & ST_OFF_GRID)
        return fail(ctx, PS_ERR_OFF_GRID, "fp32 sample is not an integer multiple of quantum (or |count| >= 2^23)");
    if (st & ST_VERIFY_MISMATCH) {
        const SmallLayout *h = ctx->h_small.as<SmallLayout>();
        return fail(ctx, PS_ERR_INTERNAL, "verify mode: fp32 screen disagreed with the exact scan (window [%lld,%lld): screen %lld, exact %lld)",
                    static_cast<long long>(h->stamp[4]), static_cast<long long>(h->stamp[5]), static_cast<long long>(h->stamp[6]), static_cast<long long>(h->stamp[7]));
    }
    if (st & ST_STACK_OVERFLOW) return fail(ctx, PS_ERR_INTERNAL, "device DFS stack overflow");
    if (st & ST_OUT_OVERFLOW) return fail(ctx, PS_ERR_INTERNAL, "device scratch overflow");
    return PS_OK;
}


// Runs spine_kernel over `jobs`, returns the per-tile anchor lists.
int run_spines(ps_ctx *ctx, const DevCfg &cfg, const std::vector<SpineJob> &jobs, int64_t scratch_entries,
               std::vector<TileList> &out)
{
    const size_t nj = jobs.size();
    out.assign(nj, TileList());
    if (nj == 0) return PS_OK;
    HIP_TRY(ctx, ctx->spine_jobs.reserve(nj * sizeof(SpineJob)));
    HIP_TRY(ctx, ctx->spine_scratch.reserve(static_cast<size_t>(scratch_entries) * sizeof(int2)));
    HIP_TRY(ctx, ctx->spine_dense.reserve(static_cast<size_t>(scratch_entries) * sizeof(int2)));
    HIP_TRY(ctx, ctx->spine_meta.reserve(nj * sizeof(int4)));
    HIP_TRY(ctx, ctx->h_up.reserve(nj * sizeof(SpineJob)));
    std::memcpy(ctx->h_up.p, jobs.data(), nj * sizeof(SpineJob));
    SmallLayout *sm = ctx->small.as<SmallLayout>();
    HIP_TRY(ctx, hipMemcpyAsync(ctx->spine_jobs.p, ctx->h_up.p, nj * sizeof(SpineJob), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(&sm->dense, 0, sizeof(unsigned long long), ctx->stream));
    if (ctx->timing >= 2) HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    {
        const unsigned g = static_cast<unsigned>(nj);
        const bool f32 = cfg.dtype == PS_DTYPE_F32;
        int lrc = ctx->spine_nt == 256
                      ? (f32 ? launch_spine<256, PS_DTYPE_F32>(ctx, cfg, g, sm) : launch_spine<256, PS_DTYPE_I16>(ctx, cfg, g, sm))
                  : ctx->spine_nt == 512
                      ? (f32 ? launch_spine<512, PS_DTYPE_F32>(ctx, cfg, g, sm) : launch_spine<512, PS_DTYPE_I16>(ctx, cfg, g, sm))
                      : (f32 ? launch_spine<1024, PS_DTYPE_F32>(ctx, cfg, g, sm) : launch_spine<1024, PS_DTYPE_I16>(ctx, cfg, g, sm));
        if (lrc) return lrc;
    }
    if (ctx->timing >= 2) HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    HIP_TRY(ctx, ctx->h_meta.reserve(nj * sizeof(int4)));
    HIP_TRY(ctx, ctx->h_small.reserve(sizeof(SmallLayout)));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_meta.p, ctx->spine_meta.p, nj * sizeof(int4), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_small.p, ctx->small.p, sizeof(SmallLayout), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const SmallLayout hs = *ctx->h_small.as<SmallLayout>();
    int rc = check_status(ctx, static_cast<unsigned>(hs.status));
    if (rc) return rc;
    const size_t total = static_cast<size_t>(hs.dense);
    if (total) {
        HIP_TRY(ctx, ctx->h_dense.reserve(total * sizeof(int2)));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_dense.p, ctx->spine_dense.p, total * sizeof(int2), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    float ms = 0;
    if (ctx->timing >= 2 && hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]) == hipSuccess) ctx->ms[0] += ms;
    const int4 *meta = ctx->h_meta.as<int4>();
    const int2 *dense = ctx->h_dense.as<int2>();
    for (size_t j = 0; j < nj; ++j) {
        TileList &t = out[j];
        t.start = jobs[j].start;
        t.ended = meta[j].y != 0;
        t.a.resize(static_cast<size_t>(meta[j].x));
        for (int i = 0; i < meta[j].x; ++i) {
            t.a[i].pos = dense[meta[j].z + i].x;
            t.a[i].kind = dense[meta[j].z + i].y;
        }
    }
    return PS_OK;
}

// anchors of one chain are >= min_width apart and at most one lies at or beyond `stop`
inline int64_t spine_cap(int64_t start, int64_t stop, int mw) { return (stop - start) / mw + 4; }

// position of `pos` in a tile list (exact match), -1 if the tile starts there, -2 if absent
int find_in(const TileList &t, int32_t pos)
{
    if (pos == t.start) return -1;
    auto it = std::lower_bound(t.a.begin(), t.a.end(), pos, [](const Anchor &x, int32_t v) { return x.pos < v; });
    if (it != t.a.end() && it->pos == pos) return static_cast<int>(it - t.a.begin());
    return -2;
}

}  // namespace

namespace {
constexpr int RC_FALLBACK = 1;      // internal: the device stitch gave up, use the host-stitch pipeline
constexpr int RC_WIDE = 2;          // internal: counts too wide for the block sums, use the LDS-window scan
// Phase 3 onwards.  Expects tree_jobs / items / first_item / ev_off populated on the device.
// d_hdr == nullptr: n_tj / n_items are the exact counts (host stitch).  Otherwise they are upper bounds used
// to size the launches and the kernels read the counts from *d_hdr on the device; the header and the status
// word are then checked after the single synchronisation at the end (*hdr_out receives the header).
int finish_batch(ps_ctx *ctx, const DevCfg &cfg, size_t n_tj, int64_t n_items, int32_t n_ev,
                        int32_t *d_bounds, int64_t cap, int64_t *h_bounds_off, ps_segstat *d_stats,
                        std::chrono::steady_clock::time_point t_begin, const AsmHeader *d_hdr = nullptr,
                        AsmHeader *hdr_out = nullptr, bool wide_check = false)
{
    int rc;
    const size_t evb = (static_cast<size_t>(n_ev) + 1) * sizeof(int64_t);
    SmallLayout *sm = ctx->small.as<SmallLayout>();
    if (ctx->timing >= 2) HIP_TRY(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    if (n_tj) {
        const unsigned g = static_cast<unsigned>(std::min<size_t>(n_tj, 0x7fffffff));
        const bool f32 = cfg.dtype == PS_DTYPE_F32;
        int lrc = cfg.bsum != nullptr && cfg.bs_wide && ctx->tree_par
                      ? (f32 ? launch_tree_par<PS_DTYPE_F32 | DT_WIDE>(ctx, cfg, g, sm, n_tj, d_hdr) : launch_tree_par<PS_DTYPE_I16 | DT_WIDE>(ctx, cfg, g, sm, n_tj, d_hdr))
                  : cfg.bsum != nullptr && cfg.bs_wide && ctx->tree_mw
                      ? (f32 ? launch_tree_mw<PS_DTYPE_F32 | DT_WIDE>(ctx, cfg, g, sm, n_tj, d_hdr) : launch_tree_mw<PS_DTYPE_I16 | DT_WIDE>(ctx, cfg, g, sm, n_tj, d_hdr))
                  : cfg.bsum != nullptr && cfg.bs_wide
                      ? (f32 ? launch_tree<64, PS_DTYPE_F32 | DT_WIDE>(ctx, cfg, g, sm, n_tj, d_hdr) : launch_tree<64, PS_DTYPE_I16 | DT_WIDE>(ctx, cfg, g, sm, n_tj, d_hdr))
                  : cfg.bsum != nullptr && ctx->tree_mw
                      ? (f32 ? launch_tree_mw<PS_DTYPE_F32>(ctx, cfg, g, sm, n_tj, d_hdr) : launch_tree_mw<PS_DTYPE_I16>(ctx, cfg, g, sm, n_tj, d_hdr))
                  : cfg.bsum != nullptr
                      ? (f32 ? launch_tree<64, PS_DTYPE_F32>(ctx, cfg, g, sm, n_tj, d_hdr) : launch_tree<64, PS_DTYPE_I16>(ctx, cfg, g, sm, n_tj, d_hdr))
                  : ctx->tree_nt == 512
                      ? (f32 ? launch_tree<512, PS_DTYPE_F32>(ctx, cfg, g, sm, n_tj, d_hdr) : launch_tree<512, PS_DTYPE_I16>(ctx, cfg, g, sm, n_tj, d_hdr))
                      : (f32 ? launch_tree<256, PS_DTYPE_F32>(ctx, cfg, g, sm, n_tj, d_hdr) : launch_tree<256, PS_DTYPE_I16>(ctx, cfg, g, sm, n_tj, d_hdr));
        if (lrc) return lrc;
    }
    if (ctx->timing >= 2) HIP_TRY(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    const bool fused = d_hdr != nullptr && ctx->gather_fused;      // (device stitch: job index == item index)
    if (fused) {
        const int64_t blocks = (n_items >> GS_LOG) + 1;
        const unsigned gg = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(blocks, 1024)));
        hipLaunchKernelGGL(gather_scan_kernel, dim3(gg), dim3(1 << GS_LOG), 0, ctx->stream, ctx->items.as<Item>(),
                           ctx->tree_jobs.as<TreeJob>(), ctx->tree_counts.as<int32_t>(), ctx->tree_scratch.as<int32_t>(),
                           static_cast<long long>(n_items), d_bounds, cap, ctx->d_is_spine, d_hdr, ctx->first_item.as<int64_t>(),
                           n_ev, ctx->bounds_off.as<int64_t>());
        HIP_TRY(ctx, hipGetLastError());
    } else {
    hipLaunchKernelGGL(item_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->items.as<Item>(),
                       ctx->tree_counts.as<int32_t>(), n_items, ctx->item_pos.as<int64_t>(), d_hdr,
                       ctx->first_item.as<int64_t>(), n_ev, ctx->bounds_off.as<int64_t>());
    HIP_TRY(ctx, hipGetLastError());
    }
    if (n_items && !fused) {
        const unsigned gg = static_cast<unsigned>(d_hdr ? std::min<int64_t>(n_items, 16384) : n_items);
        hipLaunchKernelGGL(gather_kernel, dim3(gg), dim3(64), 0, ctx->stream,
                           ctx->items.as<Item>(), ctx->tree_jobs.as<TreeJob>(), ctx->tree_counts.as<int32_t>(),
                           ctx->tree_scratch.as<int32_t>(), ctx->item_pos.as<int64_t>(), n_items, d_bounds, cap, ctx->d_is_spine,
                           d_hdr);
        HIP_TRY(ctx, hipGetLastError());
    }
    const bool stats_from_digest = d_stats != nullptr && cfg.bsum != nullptr && cfg.blk_mm != nullptr;
    if (stats_from_digest) {
        // K2 from the K0 digest: launched before the sync, the segment count is read on the device
        const int64_t scap = cap + n_ev;
        const unsigned sg = static_cast<unsigned>(std::min<int64_t>(std::max<int64_t>(scap, 1), 8192));
        if (cfg.dtype == PS_DTYPE_F32)
            hipLaunchKernelGGL(segstat_bs_kernel<PS_DTYPE_F32>, dim3(sg), dim3(64), 0, ctx->stream, cfg, ctx->ev_off.as<int64_t>(),
                               ctx->ev_len.as<int64_t>(), n_ev, d_bounds, ctx->bounds_off.as<int64_t>(), d_stats, scap,
                               reinterpret_cast<unsigned *>(&sm->status), d_hdr);
        else
            hipLaunchKernelGGL(segstat_bs_kernel<PS_DTYPE_I16>, dim3(sg), dim3(64), 0, ctx->stream, cfg, ctx->ev_off.as<int64_t>(),
                               ctx->ev_len.as<int64_t>(), n_ev, d_bounds, ctx->bounds_off.as<int64_t>(), d_stats, scap,
                               reinterpret_cast<unsigned *>(&sm->status), d_hdr);
        HIP_TRY(ctx, hipGetLastError());
        if (ctx->timing >= 2) HIP_TRY(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
    }
    const bool one_copy = ctx->bounds_off.p == ctx->small.as<char>() + sizeof(SmallLayout);
    if (one_copy) {
        HIP_TRY(ctx, ctx->h_small.reserve(sizeof(SmallLayout) + evb));
        void *h_dev = nullptr;                         // the pinned block as the device sees it (else: plain copy)
        if (ctx->download_by_kernel && hipHostGetDevicePointer(&h_dev, ctx->h_small.p, 0) != hipSuccess) { h_dev = nullptr; (void)hipGetLastError(); }
        if (h_dev) {
            static_assert(sizeof(SmallLayout) % 8 == 0, "the status block is copied in 8-byte words");
            const long long nw = static_cast<long long>((sizeof(SmallLayout) + evb) / 8);
            hipLaunchKernelGGL(download_kernel, dim3(static_cast<unsigned>(std::min<long long>((nw + 255) / 256, 64))), dim3(256), 0, ctx->stream,
                               ctx->small.as<unsigned long long>(), static_cast<unsigned long long *>(h_dev), nw);
            HIP_TRY(ctx, hipGetLastError());
        } else {
            HIP_TRY(ctx, hipMemcpyAsync(ctx->h_small.p, ctx->small.p, sizeof(SmallLayout) + evb, hipMemcpyDeviceToHost, ctx->stream));
        }
    } else {
        HIP_TRY(ctx, ctx->h_meta.reserve(evb));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_meta.p, ctx->bounds_off.p, evb, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_small.p, ctx->small.p, sizeof(SmallLayout), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (ctx->timing >= 1) HIP_TRY(ctx, hipEventRecord(ctx->ev[9], ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    {
        float seq = 0;                                 // the call's device work, upload to result copies, by HIP events
        if (ctx->timing >= 1 && hipEventElapsedTime(&seq, ctx->ev[8], ctx->ev[9]) == hipSuccess) ctx->ms[7] = seq;
    }
    std::memcpy(h_bounds_off, one_copy ? ctx->h_small.as<char>() + sizeof(SmallLayout) : ctx->h_meta.as<char>(), evb);
    const SmallLayout hs = *ctx->h_small.as<SmallLayout>();
    if (wide_check && (static_cast<unsigned>(hs.status) & ST_WIDE_RANGE)) return RC_WIDE;    // counts too wide for the block sums
    rc = check_status(ctx, static_cast<unsigned>(hs.status));
    if (rc) return rc;
    if (d_hdr) {
        const AsmHeader hd = hs.hdr;
        if (hdr_out) *hdr_out = hd;
        if (hd.fail) {
            if (ctx->debug) {                          // which seams gave up (option debug)
                const size_t nt = static_cast<size_t>(ctx->counters[2]);
                std::vector<int4> bm(nt), mt(nt);
                std::vector<SpineJob> jb(nt);
                if (nt && hipMemcpy(bm.data(), ctx->bmeta.p, nt * sizeof(int4), hipMemcpyDeviceToHost) == hipSuccess &&
                    hipMemcpy(mt.data(), ctx->spine_meta.p, nt * sizeof(int4), hipMemcpyDeviceToHost) == hipSuccess &&
                    hipMemcpy(jb.data(), ctx->spine_jobs.p, nt * sizeof(SpineJob), hipMemcpyDeviceToHost) == hipSuccess) {
                    int shown = 0;
                    size_t n_open = 0, n_gave_up = 0, n_defer = 0;
                    for (size_t g = 0; g < nt; ++g) {
                        if (bm[g].w == 4) ++n_defer;
                        else if (bm[g].w == 3 && mt[g].x == 0 && bm[g].x == 0) ++n_open;
                        else if (bm[g].w == 3) ++n_gave_up;
                    }
                    fprintf(stderr, "[poreseg] stitch failed: %zu tiles; bridges that gave up after anchors of their own %zu, open tiles not bridged %zu, deferred and not finished %zu\n",
                            nt, n_gave_up, n_open, n_defer);
                    for (size_t g = 0; g < nt && shown < 6; ++g)
                        if ((bm[g].w == 3 && !(mt[g].x == 0 && bm[g].x == 0)) || bm[g].w == 4) {
                            fprintf(stderr, "[poreseg] stitch failed: tile %zu (event %d, tile %zu of %d, [%d, %d) of %d) bridge status %d after %d anchors (next window %d); its own list: %d anchors, ended %d, open at %d\n",
                                    g, jb[g].ev, g - static_cast<size_t>(jb[g].first_tile), jb[g].ntiles, jb[g].start, jb[g].stop, jb[g].end,
                                    bm[g].w, bm[g].x, bm[g].y, mt[g].x, mt[g].y, mt[g].z);
                            ++shown;
                        }
                }
            }
            return RC_FALLBACK;
        }
    }
    ctx->counters[0] = static_cast<int64_t>(hs.work0);
    ctx->counters[1] = static_cast<int64_t>(hs.work1);
    ctx->counters[5] = static_cast<int64_t>(hs.work2 & 0xffffffffULL) + static_cast<int64_t>(hs.work2 >> 32);   // exact decisions
    ctx->counters[6] = static_cast<int64_t>(hs.work2 >> 32);                                                   // of which full fp64 window scans (block-sum scan)
#ifndef PS_STAMP
    for (int k = 0; k < 3; ++k) ctx->counters[8 + k] = static_cast<int64_t>(hs.life[3 * k + 1]);              // window scans of the spine / bridge / subtree kernels
    // near-tie decisions (seg_bs.hpp: bs_decide); -1: not counted -- the call ran on the LDS-window kernels, which decide every window
    // by one fp64 scan and keep no margins (a caller that redoes near ties on the exact route redoes such a call)
    ctx->counters[11] = cfg.bsum ? static_cast<int64_t>(hs.stamp[0]) : -1;
#endif
    ctx->counters[12] = static_cast<int64_t>(hs.lat_ctl[4]);                                                   // look-ahead helpers: chunk results published
    ctx->counters[13] = static_cast<int64_t>(hs.lat_ctl[5]);                                                   // ... and taken by an owner instead of scanning
#ifdef PS_STAMP
    {
        static const char *nm[12] = {"level", "sweep", "drain", "decide", "contend", "setup", "exact", "outside", "-", "-", "-", "-"};
        unsigned long long tot = 0;
        for (int i = 0; i < 8; ++i) tot += hs.stamp[i];
        fprintf(stderr, "[poreseg stamps] lane-0 cycles summed over waves (spine+bridge+tree):");
        for (int i = 0; i < 8; ++i) fprintf(stderr, " %s=%.1f%%", nm[i], 100.0 * hs.stamp[i] / (tot ? tot : 1));
        fprintf(stderr, " total=%llu cycles, windows=%llu -> %.0f cycles/window\n", tot, hs.work0, (double)tot / (hs.work0 ? hs.work0 : 1));
        fprintf(stderr, "[poreseg stamps] rows swept %llu, non-empty drains %llu, blocks drained %llu, hit-like windows %llu\n",
                hs.stamp[8], hs.stamp[9], hs.stamp[10], hs.stamp[11]);
        fprintf(stderr, "[poreseg stamps] blocks queued although not prunable (first block of a window, variance floor): %llu\n", hs.stamp[6]);
        for (int k = 0; k < 3; ++k)
            fprintf(stderr, "[poreseg stamps] %s: longest workgroup %llu cycles, windows %llu, sum of lifetimes %llu cycles\n",
                    k == 0 ? "spine" : k == 1 ? "bridge" : "tree", hs.life[3 * k], hs.life[3 * k + 1], hs.life[3 * k + 2]);
    }
#endif
    const int64_t total = h_bounds_off[n_ev];
    if (total > cap)
        return fail(ctx, PS_ERR_CAPACITY, "bounds capacity %lld < required %lld", static_cast<long long>(cap),
                    static_cast<long long>(total));
    if (d_stats && !stats_from_digest) {
        const int64_t nseg = total + n_ev;
        if (nseg > 0) {
            if (cfg.dtype == PS_DTYPE_F32)
                hipLaunchKernelGGL(segstat_kernel<PS_DTYPE_F32>, dim3(static_cast<unsigned>(nseg)), dim3(STAT_NT), 0,
                                   ctx->stream, cfg, ctx->ev_off.as<int64_t>(), ctx->ev_len.as<int64_t>(), n_ev, d_bounds,
                                   ctx->bounds_off.as<int64_t>(), d_stats, reinterpret_cast<unsigned *>(&sm->status));
            else
                hipLaunchKernelGGL(segstat_kernel<PS_DTYPE_I16>, dim3(static_cast<unsigned>(nseg)), dim3(STAT_NT), 0,
                                   ctx->stream, cfg, ctx->ev_off.as<int64_t>(), ctx->ev_len.as<int64_t>(), n_ev, d_bounds,
                                   ctx->bounds_off.as<int64_t>(), d_stats, reinterpret_cast<unsigned *>(&sm->status));
            HIP_TRY(ctx, hipGetLastError());
        }
    }
    if (!stats_from_digest) {
        if (ctx->timing >= 2) HIP_TRY(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    float ms = 0;
    if (ctx->timing >= 2 && hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]) == hipSuccess) ctx->ms[1] = ms;
    if (ctx->timing >= 2 && hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4]) == hipSuccess) ctx->ms[2] = ms;
    ctx->ms[3] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return PS_OK;
}



// The helpers' shared state for this call (none when switched off, or not the block-sum pipeline)
LatHelp lat_help_of(ps_ctx *ctx, const DevCfg &cfg, SmallLayout *sm)
{
    if (!ctx->lat_help || cfg.bsum == nullptr || !ctx->lat_res.p) return LAT_NONE;
    LatHelp h;
    h.ctl = sm->lat_ctl;
    h.state = ctx->lat_state.as<unsigned long long>();
    h.seam = ctx->lat_seam.as<int>();
    h.prog = sm->lat_prog;
    h.res = ctx->lat_res.as<unsigned long long>();
    h.tag_base = ctx->lat_tag_base;
    h.stay = ctx->lat_help == 2 ? 1 : 0;
    return h;
}

template <int DT> int launch_bridge_la(ps_ctx *ctx, const DevCfg &cfg, unsigned nj, SmallLayout *sm)
{
    const unsigned grid = std::min(nj, resident_slots(ctx, bridge_la_kernel<DT>, 64 * BR_LA, 0));
    hipLaunchKernelGGL((bridge_la_kernel<DT>), dim3(grid), dim3(64 * BR_LA), 0, ctx->stream, cfg,
                       ctx->spine_jobs.as<SpineJob>(), ctx->spine_scratch.as<int2>(), ctx->spine_meta.as<int4>(),
                       ctx->bridges.as<int2>(), ctx->bmeta.as<int4>(), reinterpret_cast<unsigned *>(&sm->status),
                       &sm->work0, static_cast<int>(nj), nullptr, nullptr, nullptr, 0, ctx->bridge_budget, static_cast<int>(EXT_MAX),
                       lat_help_of(ctx, cfg, sm));
    HIP_TRY(ctx, hipGetLastError());
    return PS_OK;
}

template <int NT, int DT> int launch_bridge(ps_ctx *ctx, const DevCfg &cfg, unsigned nj, SmallLayout *sm)
{
    const size_t lds = NT == 64 ? static_cast<size_t>(ctx->scan_lds_pad) : lds_bytes_for(cfg.lds_cap, NT);
    HIP_TRY(ctx, set_dyn_lds(ctx, reinterpret_cast<const void *>(bridge_kernel<NT, DT>), static_cast<int>(lds)));
    const unsigned grid = std::min(nj, resident_slots(ctx, bridge_kernel<NT, DT>, NT, lds));
    hipLaunchKernelGGL((bridge_kernel<NT, DT>), dim3(grid), dim3(NT), lds, ctx->stream, cfg,
                       ctx->spine_jobs.as<SpineJob>(), ctx->spine_scratch.as<int2>(), ctx->spine_meta.as<int4>(),
                       ctx->bridges.as<int2>(), ctx->bmeta.as<int4>(), reinterpret_cast<unsigned *>(&sm->status),
                       &sm->work0, static_cast<int>(nj), ctx->bridge_single, ctx->bridge_budget,
                       NT == 64 ? lat_help_of(ctx, cfg, sm) : LAT_NONE);
    HIP_TRY(ctx, hipGetLastError());
    return PS_OK;
}

// Device-stitch pipeline: tile spines without halo, seam bridges, assemble kernel (true spine,
// tree jobs, items) -- no host round trip before the final synchronisation.
// Returns RC_FALLBACK when a seam could not be bridged on the device (rare); the caller then runs
// the host-stitch pipeline, which repairs seams one by one.
// digest_ready (round 6, ps_detect_segment_trace): the digest of the WHOLE trace is already in the context's buffers (K0 ran over
// it as one event, trace-aligned blocks): no K0 here; the events get (centre of the trace, phase, first block) from
// ev_info_trace_kernel and the scans run in shifted coordinates (seg_bs.hpp: scan_window_ph).
int device_stitch_batch_(ps_ctx *ctx, const DevCfg &cfg_in, int bs_mode, const int64_t *ev_start, const int64_t *ev_len, int32_t n_ev, int mw, int W,
                        int32_t *d_bounds, int64_t cap, int64_t *h_bounds_off, ps_segstat *d_stats,
                        std::chrono::steady_clock::time_point t_begin, bool digest_ready = false)
{
    DevCfg cfg = cfg_in;
    const bool use_bs = bs_mode != 0;                  // 1: block-sum scan on the 32-bit digest, 2: on the 64-bit (wide) digest
    const bool wide = bs_mode == 2;
    cfg.bs_wide = wide ? 1 : 0;
    int64_t L = ctx->tile_len;
    // The tile tables depend only on the event layout and the parameters: a call that repeats the previous one's
    // (the bench loop; a file segmented with several parameter sets) reuses the tables that are still on the device.
    ps_ctx::TileCache &tc = ctx->tile_cache;
    const size_t evb = (static_cast<size_t>(n_ev) + 1) * sizeof(int64_t);
    bool reuse = tc.valid && tc.n_ev == n_ev && tc.mw == mw && tc.W == W && tc.L == L && tc.use_bs == use_bs &&
                 std::equal(ev_start, ev_start + n_ev, tc.ev_start.begin()) && std::equal(ev_len, ev_len + n_ev, tc.ev_len.begin());
    if (!reuse) {
        tc.valid = false;
        int64_t total_len = 0, sample_end = 0;
        for (int e = 0; e < n_ev; ++e) { total_len += ev_len[e]; sample_end = std::max(sample_end, ev_start[e] + ev_len[e]); }
        int64_t Lt = L;
        // block-sum scan: one tile per wave slot of the scan kernels (4 096 at four waves per SIMD) when the tiles stay
        // long (>= 8 W: a 1e9-sample trace, 3.05 -> 2.70 ms), else 1 536 tiles of at least 4 W (shorter tiles only add
        // speculative windows and seams: measured on the 1e8-sample trace; round 4, with the lighter windows
        // of the coarse pass: 2 048 -> 1 536 tiles, 0.2298 -> 0.2247 ms per step in five interleaved rounds, one call unchanged)
        if (Lt <= 0) Lt = use_bs ? ((total_len + 4095) / 4096 >= 8LL * W ? (total_len + 4095) / 4096
                                                                         : std::max<int64_t>(4LL * W, (total_len + 1535) / 1536))
                                 : std::max<int64_t>(8LL * W, (total_len + 1023) / 1024);
        Lt = (Lt + 7) & ~7LL;
        Lt = std::min<int64_t>(Lt, 0x7fffffff);
        std::vector<SpineJob> jobs;
        std::vector<int64_t> ev_first_tile(static_cast<size_t>(n_ev) + 1, 0);
        int64_t list_entries = 0, vb_run = 0;
        for (int e = 0; e < n_ev; ++e) {
            ev_first_tile[e] = static_cast<int64_t>(jobs.size());
            const int64_t len = ev_len[e];
            if (len == 0) continue;
            const int64_t vbase = vb_run;
            vb_run += len;
            // tiles of one event share one length: the event is cut evenly.  An event of up to 1.5 tile lengths stays ONE tile
            // (round 6: a 50k event with L = 40k used to be 2 x 25k -- two chains of three windows, a seam, a bridge and a stitch
            // entry for nothing: every event start is a true anchor.  BASELINE config 2, 1 024 x 50 000: 0.338-0.355 -> 0.309 ms
            // for a lone call, 2 048 -> 1 024 tiles, 15 459 -> 14 435 windows; profiles/r06_experiments/r6_config2_tile.txt)
            const int64_t nt0 = use_bs ? std::max<int64_t>(1, (len + Lt / 2) / Lt) : (len + Lt - 1) / Lt;
            const int64_t Le = std::min<int64_t>(((len + nt0 - 1) / nt0 + 7) & ~7LL, 0x7fffffff);
            const int64_t nt = (len + Le - 1) / Le;
            if (static_cast<int64_t>(jobs.size()) + nt > 0x7ffffff0) return RC_FALLBACK;
            for (int64_t t = 0; t < nt; ++t) {
                SpineJob j;
                j.base = ev_start[e];
                j.start = static_cast<int32_t>(t * Le);
                j.end = static_cast<int32_t>(len);
                j.stop = static_cast<int32_t>(t == nt - 1 ? len : (t + 1) * Le);
                j.out_cap = static_cast<int32_t>(std::min<int64_t>((j.stop - j.start) / mw + 4, 0x7fffffff));
                j.out_off = list_entries;
                j.first_tile = static_cast<int32_t>(ev_first_tile[e]);
                j.ntiles = static_cast<int32_t>(nt);
                j.tile_len = static_cast<int32_t>(Le);
                j.ev = e;
                j.vbase = vbase;
                list_entries += j.out_cap;
                jobs.push_back(j);
            }
        }
        ev_first_tile[n_ev] = static_cast<int64_t>(jobs.size());
        const size_t nj = jobs.size();
        // one upload: [jobs | ev_first_tile | ev_start | ev_len | ev_boff] -> one device blob the five arrays point into
        const size_t jb = (nj * sizeof(SpineJob) + 15) & ~static_cast<size_t>(15);
        std::vector<int64_t> boff(static_cast<size_t>(n_ev) + 1, 0);           // first block of every event (K0)
        for (int e = 0; e < n_ev; ++e) boff[e + 1] = boff[e] + (ev_len[e] + 7) / 8;
        const size_t up_bytes = jb + 4 * evb;
        HIP_TRY(ctx, ctx->h_up.reserve(up_bytes + 64));
        HIP_TRY(ctx, ctx->up_dev.reserve(up_bytes + 64));
        char *up = ctx->h_up.as<char>();
        if (nj) std::memcpy(up, jobs.data(), nj * sizeof(SpineJob));
        std::memcpy(up + jb, ev_first_tile.data(), evb);
        std::memcpy(up + jb + evb, ev_start, evb - sizeof(int64_t));
        std::memcpy(up + jb + 2 * evb, ev_len, evb - sizeof(int64_t));
        std::memcpy(up + jb + 3 * evb, boff.data(), evb);
        tc.n_ev = n_ev; tc.mw = mw; tc.W = W; tc.L = L; tc.use_bs = use_bs;
        tc.ev_start.assign(ev_start, ev_start + n_ev); tc.ev_len.assign(ev_len, ev_len + n_ev);
        tc.nj = nj; tc.jb = jb; tc.up_bytes = up_bytes; tc.list_entries = list_entries; tc.total_len = total_len;
        tc.sample_end = sample_end; tc.nb_total = boff[n_ev];
    }
    const size_t nj = tc.nj, jb = tc.jb, up_bytes = tc.up_bytes;
    const int64_t list_entries = tc.list_entries, total_len = tc.total_len, sample_end = tc.sample_end;
    ctx->counters[2] = static_cast<int64_t>(nj);
    const int64_t max_items = list_entries + static_cast<int64_t>(nj) * BR_MAX;
    const int64_t tscratch_bound = total_len / mw + max_items + 1;     // tree output regions: (base+pred)/mw + item

    HIP_TRY(ctx, ctx->spine_scratch.reserve(std::max<int64_t>(1, list_entries) * sizeof(int2)));
    HIP_TRY(ctx, ctx->spine_meta.reserve(std::max<size_t>(4, nj) * sizeof(int4)));
    HIP_TRY(ctx, ctx->bridges.reserve(std::max<size_t>(1, nj) * BR_MAX * sizeof(int2)));
    if (use_bs && ctx->lat_help) {
        // (published chunk results carry the tag of their listing; a fresh buffer is zero, tags start at 1, and before the 24
        //  bits run out the buffer is cleared)
        const size_t res_bytes = static_cast<size_t>(LAT_D) * LAT_C * sizeof(unsigned long long);
        HIP_TRY(ctx, ctx->lat_state.reserve(LAT_D * sizeof(unsigned long long)));
        HIP_TRY(ctx, ctx->lat_seam.reserve(LAT_D * sizeof(int)));
        HIP_TRY(ctx, ctx->lat_res.reserve_zeroed(res_bytes, ctx->stream));
        if (ctx->lat_tag_next + 2u * LAT_TAGS >= (1u << 24)) {
            HIP_TRY(ctx, hipMemsetAsync(ctx->lat_res.p, 0, res_bytes, ctx->stream));
            ctx->stream_idle = false;
            ctx->lat_tag_next = 1;
        }
        ctx->lat_tag_base = ctx->lat_tag_next;
        ctx->lat_tag_next += LAT_TAGS;
    }
    HIP_TRY(ctx, ctx->bmeta.reserve(std::max<size_t>(1, nj) * sizeof(int4)));
    HIP_TRY(ctx, ctx->tile_i32.reserve(std::max<size_t>(1, nj) * 4 * sizeof(int)));
    HIP_TRY(ctx, ctx->sp_off.reserve((nj + 1) * sizeof(long long)));
    HIP_TRY(ctx, ctx->spine_items.reserve(std::max<int64_t>(1, max_items) * sizeof(int4)));
    HIP_TRY(ctx, ctx->tree_jobs.reserve(std::max<int64_t>(1, max_items) * sizeof(TreeJob)));
    HIP_TRY(ctx, ctx->tree_counts.reserve((std::max<int64_t>(1, max_items) + (max_items >> GS_LOG) + 2) * sizeof(int32_t)));   // (+ the sums per 256 jobs: gather_scan_kernel)
    HIP_TRY(ctx, ctx->items.reserve(std::max<int64_t>(1, max_items) * sizeof(Item)));
    HIP_TRY(ctx, ctx->item_pos.reserve((static_cast<size_t>(max_items) + 1) * sizeof(int64_t)));
    HIP_TRY(ctx, ctx->tree_scratch.reserve(static_cast<size_t>(tscratch_bound) * sizeof(int32_t)));
    HIP_TRY(ctx, ctx->tree_spill.reserve(static_cast<size_t>(tscratch_bound) * sizeof(int2)));
    HIP_TRY(ctx, ctx->first_item.reserve(evb));
    if (evb <= SMALL_TAIL) ctx->bounds_off.alias(ctx->small.as<char>() + sizeof(SmallLayout));   // comes back with the status block
    else HIP_TRY(ctx, ctx->bounds_off.reserve(evb));
    ctx->asm_hdr.alias(&ctx->small.as<SmallLayout>()->hdr);
    HIP_TRY(ctx, ctx->h_small.reserve(sizeof(SmallLayout)));
    // Where upload + K0 are queued: the context's own stream, or the device's front stream (k0_shared) -- then under the front
    // stream's mutex until K0's event is recorded, and the context's stream continues behind that event.
    FrontStream *front = (ctx->k0_shared && use_bs && tc.nj) ? front_for(ctx) : nullptr;
    hipStream_t fs = front ? front->s : ctx->stream;
    std::unique_lock<std::mutex> front_lock;
    if (front) {
        front_lock = std::unique_lock<std::mutex>(front->mu);
        if (!ctx->stream_idle) {                       // (work the caller left on the context's stream -- a filter, a re-quantisation -- comes first)
            HIP_TRY(ctx, hipEventRecord(ctx->ev_front[0], ctx->stream));
            HIP_TRY(ctx, hipStreamWaitEvent(fs, ctx->ev_front[0], 0));
        }
    }
    {
        const char *up = ctx->h_up.as<char>();
        void *up_devptr = nullptr;                     // the pinned blob as the device sees it (else: plain copy)
        if (ctx->upload_by_kernel && hipHostGetDevicePointer(&up_devptr, const_cast<char *>(up), 0) != hipSuccess) { up_devptr = nullptr; (void)hipGetLastError(); }
        if (up_devptr) {
            // (tables still on the device from the previous call: the kernel only clears the status block)
            const long long n16 = reuse ? 0 : static_cast<long long>((up_bytes + 15) / 16);
            const unsigned ug = static_cast<unsigned>(std::max<long long>(1, std::min<long long>((n16 + 255) / 256, 1024)));
            hipLaunchKernelGGL(upload_kernel, dim3(ug), dim3(256), 0, fs, static_cast<const int4 *>(up_devptr),
                               ctx->up_dev.as<int4>(), n16, ctx->small.as<unsigned long long>(),
                               static_cast<int>(sizeof(SmallLayout) / sizeof(unsigned long long)));
            HIP_TRY(ctx, hipGetLastError());
        } else {
            HIP_TRY(ctx, hipMemsetAsync(ctx->small.p, 0, sizeof(SmallLayout), fs));
            if (!reuse) HIP_TRY(ctx, hipMemcpyAsync(ctx->up_dev.p, up, up_bytes, hipMemcpyHostToDevice, fs));
        }
        tc.valid = true;
    }
    char *dup = ctx->up_dev.as<char>();
    ctx->spine_jobs.alias(dup);
    ctx->ev_first_tile.alias(dup + jb);
    ctx->ev_off.alias(dup + jb + evb);
    ctx->ev_len.alias(dup + jb + 2 * evb);
    ctx->ev_boff.alias(dup + jb + 3 * evb);

    SmallLayout *sm = ctx->small.as<SmallLayout>();
    const bool f32 = cfg.dtype == PS_DTYPE_F32;
    if (ctx->timing >= 2) HIP_TRY(ctx, hipEventRecord(ctx->ev[7], fs));
    if (use_bs && nj && digest_ready) {
        HIP_TRY(ctx, ctx->ev_info_tr.reserve(static_cast<size_t>(std::max(1, n_ev)) * sizeof(int4)));
        hipLaunchKernelGGL(ev_info_trace_kernel, dim3(static_cast<unsigned>(std::min(64, (n_ev + 255) / 256))), dim3(256), 0, fs,
                           ctx->ev_off.as<int64_t>(), n_ev, ctx->ev_info.as<int4>(), ctx->ev_info_tr.as<int4>());
        HIP_TRY(ctx, hipGetLastError());
        cfg.bsum = ctx->bsum.p;
        cfg.ev_info = ctx->ev_info_tr.as<int4>();
        cfg.chunk_tot = ctx->chunk_mabs.as<int4>();
        cfg.grp = ctx->groups ? ctx->grp.p : nullptr;
        cfg.blk_mm = d_stats ? ctx->blk_mm.as<int>() : nullptr;
    } else if (use_bs && nj) {
        // K0: chunk-prefixed block sums (one streaming pass), per-event centre m, totals + max|k| per 256 blocks
        const int64_t nb_total = tc.nb_total;
        // (a wave of K0 takes 256 blocks; the digest arrays are padded to whole waves, +1: the end boundary)
        const int64_t nb_pad = k0_padded_blocks(nb_total);
        // (at least 16 KB: a wave block of K0's general route fetches -- and ignores -- the head of this buffer, seg_bs.hpp)
        HIP_TRY(ctx, ctx->bsum.reserve(std::max<size_t>(16384, static_cast<size_t>(nb_pad) * (wide ? sizeof(int4) : sizeof(uint2)))));
        HIP_TRY(ctx, ctx->ev_info.reserve(static_cast<size_t>(std::max(1, n_ev)) * sizeof(int4)));
        HIP_TRY(ctx, ctx->chunk_mabs.reserve(static_cast<size_t>(nb_pad / BS_CHUNK + 1) * (wide ? 2 : 1) * sizeof(int4)));
        if (!wide && ctx->groups) {                    // group records of the coarse pass (16 B per 32 blocks; +1: the end boundary's)
            HIP_TRY(ctx, ctx->grp.reserve(static_cast<size_t>(nb_pad / BS_GRP + 1) * sizeof(uint4)));
            cfg.grp = ctx->grp.p;
        }
        if (d_stats && !wide) {                        // per-block min/max for the statistics kernel (4 B per block; int16 pairs)
            HIP_TRY(ctx, ctx->blk_mm.reserve(static_cast<size_t>(nb_pad) * sizeof(int)));
            cfg.blk_mm = ctx->blk_mm.as<int>();
        }
        // K0 is persistent (round 5): k0_waves workgroups per CU (K0_WAVES waves each, i.e. that many waves per SIMD), every
        // wave striding over the wave blocks of the call with its next block's samples in flight.  One or two waves per SIMD
        // saturate HBM; the rest of the SIMD stays free for the scan waves of the other calls in flight.  k0_waves = 0: one wave
        // per wave block as in rounds 3 and 4 (the launch then fills every slot the registers allow).
        if (ctx->n_cu <= 0) {
            hipDeviceProp_t prop;
            ctx->n_cu = hipGetDeviceProperties(&prop, ctx->device) == hipSuccess ? prop.multiProcessorCount : 256;
        }
        const unsigned k0_full = static_cast<unsigned>((nb_pad / K0_WB + K0_WAVES - 1) / K0_WAVES);
        // (a wave block of int16 samples is half the bytes: twice the waves keep the same bytes in flight)
        const unsigned k0_per_cu = static_cast<unsigned>(ctx->k0_waves) * ((f32 || (ctx->k0_sets > 2 && !wide)) ? 1u : 2u);   // (k0_sets > 2: diagnostic library only)
        const unsigned k0_grid = ctx->k0_waves > 0 ? std::min(k0_full, k0_per_cu * static_cast<unsigned>(ctx->n_cu) * (4u / K0_WAVES)) : k0_full;
        const size_t k0_lds = 0;
#define PS_K0N(DTV, NSV) hipLaunchKernelGGL((blocksum_kernel<DTV, NSV>), dim3(k0_grid), dim3(64 * K0_WAVES), k0_lds, fs, cfg,       \
                                    ctx->ev_off.as<int64_t>(), ctx->ev_len.as<int64_t>(), ctx->ev_boff.as<int64_t>(), n_ev, sample_end, \
                                    ctx->bsum.p, ctx->ev_info.as<int4>(), ctx->chunk_mabs.as<int4>(),                                   \
                                    reinterpret_cast<unsigned *>(&sm->status), k0_grp)
        if (ctx->k0_admit > 0 && !front) {
            const int grc = chain_enter(ctx, ctx->device, ctx->k0_admit, fs, &ctx->chain_ticket);
            if (grc) return grc;
            ctx->chain_held = true;
        }
#ifdef PS_DIAG
        const bool skip_k0 = ctx->dbg_phase == 1 && reuse;     // diagnostics: the previous call's digest
        uint4 *const k0_grp = (ctx->dbg_k0_nogrp && reuse) ? nullptr : const_cast<uint4 *>(static_cast<const uint4 *>(cfg.grp));
#else
        const bool skip_k0 = false;
        uint4 *const k0_grp = const_cast<uint4 *>(static_cast<const uint4 *>(cfg.grp));
#endif
#define PS_K0(DTV) PS_K0N(DTV, 2)
        if (skip_k0) { }
        else if (wide) { if (f32) PS_K0(PS_DTYPE_F32 | DT_WIDE); else PS_K0(PS_DTYPE_I16 | DT_WIDE); }
#ifdef PS_DIAG
        // (measured and rejected, round 6 -- docs/ROUND_6.md: fatter K0 waves hold fewer registers per byte in flight, but a lone K0
        //  wave per SIMD cannot issue fast enough beside four scan waves; the instances exist in the diagnostic library only)
        else if (ctx->k0_sets == 3 && ctx->k0_waves > 0) { if (f32) PS_K0N(PS_DTYPE_F32, 3); else PS_K0N(PS_DTYPE_I16, 6); }
        else if (ctx->k0_sets == 4 && ctx->k0_waves > 0) { if (f32) PS_K0N(PS_DTYPE_F32, 4); else PS_K0N(PS_DTYPE_I16, 8); }
#endif
        else           { if (f32) PS_K0(PS_DTYPE_F32); else PS_K0(PS_DTYPE_I16); }
#undef PS_K0
#undef PS_K0N
        HIP_TRY(ctx, hipGetLastError());
        cfg.bsum = ctx->bsum.p;
        cfg.ev_info = ctx->ev_info.as<int4>();
        cfg.chunk_tot = ctx->chunk_mabs.as<int4>();
    }
    if (ctx->timing >= 2) HIP_TRY(ctx, hipEventRecord(ctx->ev[0], fs));
    if (ctx->chain_held) {                             // K0 is behind this event: the call M tickets later waits for it on the device
        ctx->chain_held = false;
        const int crc = chain_publish(ctx, ctx->device, fs, ctx->chain_ticket);
        if (crc) return crc;
    }
#ifdef PS_DIAG
    if (ctx->dbg_phase == 2) {                           // diagnostics: K0 only
        if (front) front_lock.unlock();
        HIP_TRY(ctx, hipStreamSynchronize(fs));
        for (int e = 0; e <= n_ev; ++e) h_bounds_off[e] = 0;
        return PS_OK;
    }
#endif
    if (front) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_front[1], fs));
        front_lock.unlock();
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_front[1], 0));
    }
    if (nj && wide) {
        // (the 64-bit digest: same kernels, compiled for it)
        const unsigned g = static_cast<unsigned>(nj);
        int lrc = f32 ? launch_spine<64, PS_DTYPE_F32 | DT_WIDE>(ctx, cfg, g, sm, true) : launch_spine<64, PS_DTYPE_I16 | DT_WIDE>(ctx, cfg, g, sm, true);
        if (lrc) return lrc;
        if (ctx->timing >= 2) HIP_TRY(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
        lrc = f32 ? launch_bridge<64, PS_DTYPE_F32 | DT_WIDE>(ctx, cfg, g, sm) : launch_bridge<64, PS_DTYPE_I16 | DT_WIDE>(ctx, cfg, g, sm);
        if (lrc) return lrc;
        lrc = f32 ? launch_bridge_la<PS_DTYPE_F32 | DT_WIDE>(ctx, cfg, g, sm) : launch_bridge_la<PS_DTYPE_I16 | DT_WIDE>(ctx, cfg, g, sm);
        if (lrc) return lrc;
    } else if (nj && use_bs) {
        const unsigned g = static_cast<unsigned>(nj);
        int lrc = f32 ? launch_spine<64, PS_DTYPE_F32>(ctx, cfg, g, sm, true) : launch_spine<64, PS_DTYPE_I16>(ctx, cfg, g, sm, true);
        if (lrc) return lrc;
        if (ctx->timing >= 2) HIP_TRY(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
        // single-wave bridges first; the seams that run into a stretch without splits are finished by the look-ahead kernel
        lrc = f32 ? launch_bridge<64, PS_DTYPE_F32>(ctx, cfg, g, sm) : launch_bridge<64, PS_DTYPE_I16>(ctx, cfg, g, sm);
        if (lrc) return lrc;
        lrc = f32 ? launch_bridge_la<PS_DTYPE_F32>(ctx, cfg, g, sm) : launch_bridge_la<PS_DTYPE_I16>(ctx, cfg, g, sm);
        if (lrc) return lrc;
    } else if (nj) {
        const unsigned g = static_cast<unsigned>(nj);
        int lrc = ctx->spine_nt == 256
                      ? (f32 ? launch_spine<256, PS_DTYPE_F32>(ctx, cfg, g, sm, true) : launch_spine<256, PS_DTYPE_I16>(ctx, cfg, g, sm, true))
                  : ctx->spine_nt == 512
                      ? (f32 ? launch_spine<512, PS_DTYPE_F32>(ctx, cfg, g, sm, true) : launch_spine<512, PS_DTYPE_I16>(ctx, cfg, g, sm, true))
                      : (f32 ? launch_spine<1024, PS_DTYPE_F32>(ctx, cfg, g, sm, true) : launch_spine<1024, PS_DTYPE_I16>(ctx, cfg, g, sm, true));
        if (lrc) return lrc;
        if (ctx->timing >= 2) HIP_TRY(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
        lrc = ctx->spine_nt == 256
                  ? (f32 ? launch_bridge<256, PS_DTYPE_F32>(ctx, cfg, g, sm) : launch_bridge<256, PS_DTYPE_I16>(ctx, cfg, g, sm))
                  : (f32 ? launch_bridge<512, PS_DTYPE_F32>(ctx, cfg, g, sm) : launch_bridge<512, PS_DTYPE_I16>(ctx, cfg, g, sm));
        if (lrc) return lrc;
    }
    if (ctx->timing >= 2) HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    int *ti = ctx->tile_i32.as<int>();
    const size_t njp = std::max<size_t>(1, nj);
    // per-tile arrays of the stitch live in LDS when they fit: (nj+1) int64 + 4*nj int32
    const size_t tile_lds = (nj + 1) * sizeof(long long) + 4 * nj * sizeof(int);
    const int use_lds = tile_lds <= 150 * 1024;
    if (use_lds)
        HIP_TRY(ctx, set_dyn_lds(ctx, reinterpret_cast<const void *>(assemble_tiles_kernel), static_cast<int>(tile_lds)));
    AsmHeader hd = {};
    int rc = PS_OK;
    int64_t items_cap = max_items;                      // (grows with the seams a second-chance round extends)
    int slots_used = 0, ext_stride = EXT_MAX;
    for (int round = 0;; ++round) {
        hipLaunchKernelGGL(assemble_tiles_kernel, dim3(1), dim3(1024), use_lds ? tile_lds : 0, ctx->stream,
                           static_cast<int>(nj), ctx->spine_meta.as<int4>(), ctx->bmeta.as<int4>(),
                           ctx->ev_first_tile.as<int64_t>(), n_ev, ti, ti + njp, ti + 2 * njp, ti + 3 * njp,
                           ctx->sp_off.as<long long>(), ctx->first_item.as<int64_t>(), ctx->asm_hdr.as<AsmHeader>(),
                           static_cast<long long>(items_cap), use_lds, reinterpret_cast<const unsigned *>(&sm->status));
        HIP_TRY(ctx, hipGetLastError());
        // no host round trip here: the downstream kernels read the item count from the header on the device
        // (launches sized by the host-side upper bound), header and status are checked after the final sync
        const AsmHeader *d_hdr = ctx->asm_hdr.as<AsmHeader>();
        if (items_cap > 0) {
            const unsigned ag = static_cast<unsigned>(std::min<int64_t>((items_cap + 255) / 256, 2048));
            hipLaunchKernelGGL(assemble_items_kernel, dim3(ag), dim3(256), 0,
                               ctx->stream, ctx->spine_jobs.as<SpineJob>(), static_cast<int>(nj), ctx->spine_meta.as<int4>(),
                               ctx->spine_scratch.as<int2>(), ctx->bridges.as<int2>(), ti + 3 * njp,
                               ctx->sp_off.as<long long>(), static_cast<long long>(items_cap), mw, W, ctx->tree_jobs.as<TreeJob>(),
                               ctx->items.as<Item>(), ctx->tree_counts.as<int32_t>(), d_hdr, cfg.bsum != nullptr ? cfg.ev_info : nullptr,
                               ctx->bridge_ext.as<int2>(), ctx->ext_slot.as<int>(), ext_stride);
            HIP_TRY(ctx, hipGetLastError());
        }
        if (ctx->timing >= 2 && round == 0) HIP_TRY(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
        rc = finish_batch(ctx, cfg, static_cast<size_t>(items_cap), items_cap, n_ev, d_bounds, cap, h_bounds_off, d_stats, t_begin,
                          d_hdr, &hd, use_bs);
        // ---- second chance for seams that gave up (the block-sum pipeline only: it has the look-ahead kernel) -------------
        if (rc != RC_FALLBACK || !use_bs || !nj || round == EXT_ROUNDS || !ctx->bridge_ext_on) break;
        HIP_TRY(ctx, ctx->h_meta.reserve(2 * nj * sizeof(int4)));
        int4 *h_bm = ctx->h_meta.as<int4>(), *h_mt = h_bm + nj;
        HIP_TRY(ctx, hipMemcpyAsync(h_bm, ctx->bmeta.p, nj * sizeof(int4), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(h_mt, ctx->spine_meta.p, nj * sizeof(int4), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        // on the true path: whatever failed there (a seam out of anchors, an open tile entered at its start); elsewhere: the
        // seams that ran out of anchors (few; they may lie on the path once the first ones are mended)
        std::vector<int> list;
        bool hopeless = false;
        if (round == 0) {                              // few seams: long extensions; many (a batch of events): shorter ones
            size_t n_fail = 0;
            for (size_t g = 0; g < nj; ++g) n_fail += h_bm[g].w == BR_FAIL_REACHED || (h_bm[g].w == BR_FAIL && h_bm[g].x == ctx->bridge_budget);
            ext_stride = n_fail <= static_cast<size_t>(EXT_SLOTS) ? EXT_MAX : EXT_MAX_MANY;
        }
        const int slot_cap = EXT_SLOTS * (EXT_MAX / ext_stride);
        for (size_t g = 0; g < nj; ++g) {
            const bool reached = h_bm[g].w == BR_FAIL_REACHED;
            const bool gave_up = h_bm[g].x == ctx->bridge_budget;                 // (out of anchors in the first pass)
            if (!reached && !(h_bm[g].w == BR_FAIL && gave_up)) continue;
            const bool open_tile = h_mt[g].x == 0 && h_bm[g].x == 0;
            if (!gave_up && !open_tile) { if (reached) hopeless = true; continue; }   // (extended before and still not joined)
            if (slots_used + static_cast<int>(list.size()) < slot_cap) list.push_back(static_cast<int>(g));
            else if (reached) hopeless = true;
        }
        if (list.empty() || hopeless) break;
        if (ctx->debug) fprintf(stderr, "[poreseg] second chance, round %d: %zu seams continued on the device\n", round + 1, list.size());
        const size_t n_ext = list.size();
        HIP_TRY(ctx, ctx->bridge_ext.reserve(static_cast<size_t>(EXT_SLOTS) * EXT_MAX * sizeof(int2)));
        HIP_TRY(ctx, ctx->ext_slot.reserve(nj * sizeof(int)));
        HIP_TRY(ctx, ctx->ext_list.reserve(static_cast<size_t>(EXT_SLOTS) * (EXT_MAX / EXT_MAX_MANY) * sizeof(int)));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->ext_list.p, list.data(), n_ext * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));                       // (list is a local)
        HIP_TRY(ctx, hipMemsetAsync(&sm->hdr, 0, sizeof(AsmHeader), ctx->stream));
        {
            const bool f32 = cfg.dtype == PS_DTYPE_F32;
            const unsigned g = static_cast<unsigned>(n_ext);
#define PS_EXT(DTV) hipLaunchKernelGGL((bridge_la_kernel<DTV, true>), dim3(g), dim3(64 * BR_LA), 0, ctx->stream, cfg,                      \
                                       ctx->spine_jobs.as<SpineJob>(), ctx->spine_scratch.as<int2>(), ctx->spine_meta.as<int4>(),          \
                                       ctx->bridges.as<int2>(), ctx->bmeta.as<int4>(), reinterpret_cast<unsigned *>(&sm->status),           \
                                       &sm->work0, static_cast<int>(n_ext), ctx->ext_list.as<int>(), ctx->bridge_ext.as<int2>(),           \
                                       ctx->ext_slot.as<int>(), slots_used, BR_MAX, ext_stride)
            if (wide) { if (f32) PS_EXT(PS_DTYPE_F32 | DT_WIDE); else PS_EXT(PS_DTYPE_I16 | DT_WIDE); }
            else      { if (f32) PS_EXT(PS_DTYPE_F32); else PS_EXT(PS_DTYPE_I16); }
#undef PS_EXT
            HIP_TRY(ctx, hipGetLastError());
        }
        ctx->counters[4] += static_cast<int64_t>(n_ext);
        slots_used += static_cast<int>(n_ext);
        // room for the anchors the continued seams may add
        items_cap += static_cast<int64_t>(n_ext) * ext_stride;
        const int64_t ts2 = total_len / mw + items_cap + 1;
        HIP_TRY(ctx, ctx->spine_items.reserve(static_cast<size_t>(items_cap) * sizeof(int4)));
        HIP_TRY(ctx, ctx->tree_jobs.reserve(static_cast<size_t>(items_cap) * sizeof(TreeJob)));
        HIP_TRY(ctx, ctx->tree_counts.reserve((static_cast<size_t>(items_cap) + (static_cast<size_t>(items_cap) >> GS_LOG) + 2) * sizeof(int32_t)));
        HIP_TRY(ctx, ctx->items.reserve(static_cast<size_t>(items_cap) * sizeof(Item)));
        HIP_TRY(ctx, ctx->item_pos.reserve((static_cast<size_t>(items_cap) + 1) * sizeof(int64_t)));
        HIP_TRY(ctx, ctx->tree_scratch.reserve(static_cast<size_t>(ts2) * sizeof(int32_t)));
        HIP_TRY(ctx, ctx->tree_spill.reserve(static_cast<size_t>(ts2) * sizeof(int2)));
    }
    ctx->counters[3] = hd.n_items;
    float ms = 0;
    if (ctx->timing >= 2 && nj && hipEventElapsedTime(&ms, ctx->ev[7], ctx->ev[0]) == hipSuccess) ctx->ms[6] = ms;       // blocksum_kernel (K0)
    if (ctx->timing >= 2 && nj && hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[6]) == hipSuccess) ctx->ms[0] = ms;       // spine_kernel
    if (ctx->timing >= 2 && nj && hipEventElapsedTime(&ms, ctx->ev[6], ctx->ev[1]) == hipSuccess) ctx->ms[5] = ms;       // bridge_kernel
    if (ctx->timing >= 2 && hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[5]) == hipSuccess) ctx->ms[4] = ms;     // device stitch incl. header sync
    return rc;
}
// (whatever way the call ends: a ticket of the K0 chain that was not recorded is forgotten)
int device_stitch_batch(ps_ctx *ctx, const DevCfg &cfg_in, int bs_mode, const int64_t *ev_start, const int64_t *ev_len, int32_t n_ev, int mw, int W,
                        int32_t *d_bounds, int64_t cap, int64_t *h_bounds_off, ps_segstat *d_stats,
                        std::chrono::steady_clock::time_point t_begin, bool digest_ready = false)
{
    const int rc = device_stitch_batch_(ctx, cfg_in, bs_mode, ev_start, ev_len, n_ev, mw, W, d_bounds, cap, h_bounds_off, d_stats, t_begin, digest_ready);
    ctx->chain_held = false;
    return rc;
}
}  // namespace

namespace {
// Do 16-byte global loads from 2- and 4-byte-aligned addresses return the right bytes on this device?  (They do on gfx950 --
// tools/probes/unaligned_probe.hip, at 92 % of the aligned rate -- but the headline kernel must not rest on a stand-alone
// probe run once on one box)  Once per device and process; -1 not probed yet, 0 no, 1 yes.
std::mutex g_probe_mu;
int g_unaligned_ok[16] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
int k0_unaligned_probe(int device, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(g_probe_mu);
    int &ok = g_unaligned_ok[device & 15];
    if (ok >= 0) return ok;
    ok = 0;
    unsigned char h[128];
    for (int i = 0; i < 128; ++i) h[i] = static_cast<unsigned char>(37 * i + 11);
    unsigned char *d = nullptr;
    unsigned *bad = nullptr;
    if (hipMalloc(&d, 256) != hipSuccess || hipMalloc(&bad, sizeof(unsigned)) != hipSuccess) { if (d) (void)hipFree(d); (void)hipGetLastError(); return ok; }
    unsigned hb = 1;
    if (hipMemcpyAsync(d, h, 128, hipMemcpyHostToDevice, st) == hipSuccess && hipMemsetAsync(bad, 0, sizeof(unsigned), st) == hipSuccess) {
        hipLaunchKernelGGL(k0_unaligned_probe_kernel<2>, dim3(1), dim3(64), 0, st, d, 16, bad);     // int16: offsets 0, 2, .. 30 bytes
        hipLaunchKernelGGL(k0_unaligned_probe_kernel<4>, dim3(1), dim3(64), 0, st, d, 16, bad);     // fp32: offsets 0, 4, .. 60 bytes
        if (hipGetLastError() == hipSuccess && hipMemcpyAsync(&hb, bad, sizeof(unsigned), hipMemcpyDeviceToHost, st) == hipSuccess &&
            hipStreamSynchronize(st) == hipSuccess)
            ok = hb == 0 ? 1 : 0;
    }
    (void)hipFree(d); (void)hipFree(bad); (void)hipGetLastError();
    return ok;
}
}  // namespace

#ifdef PS_DIAG
// libporeseg_diag.so only (make -C pypore_amd/csrc diag): a new context takes its settings from PORESEG_* variables, as every
// build did until round 5.  The product library has no getenv: what a call returns depends on its arguments and on
// ps_set_option / ps_set_tiling, never on the caller's environment.
namespace {
void diag_env(ps_ctx *ctx)
{
    if (const char *e = std::getenv("PORESEG_MODE")) ctx->mode = std::atoi(e);
    if (const char *e = std::getenv("PORESEG_SPINE_NT")) ctx->spine_nt = std::atoi(e);
    if (const char *e = std::getenv("PORESEG_TREE_NT")) ctx->tree_nt = std::atoi(e);
    if (const char *e = std::getenv("PORESEG_REP_EVAL")) ctx->rep_eval = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("PORESEG_REP_STAGE")) ctx->rep_stage = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("PORESEG_REP_SUM")) ctx->rep_sum = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("PORESEG_PRUNE")) ctx->prune = std::atoi(e) != 0;
    if (const char *e = std::getenv("PORESEG_SCAN_BS")) ctx->scan_bs = std::atoi(e) != 0;
    if (const char *e = std::getenv("PORESEG_GROUPS")) ctx->groups = std::atoi(e) != 0;
    if (const char *e = std::getenv("PORESEG_TREE_PAR")) ctx->tree_par = std::atoi(e) != 0;
    if (const char *e = std::getenv("PORESEG_K0_WAVES")) ctx->k0_waves = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("PORESEG_K0_SHARED")) ctx->k0_shared = std::atoi(e) != 0;
    if (const char *e = std::getenv("PORESEG_DBG_PHASE")) ctx->dbg_phase = std::atoi(e);
    if (const char *e = std::getenv("PORESEG_SCAN_LDS_PAD")) ctx->scan_lds_pad = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("PORESEG_K0_MAX")) ctx->k0_admit = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("PORESEG_NOISE_K")) ctx->noise_k = static_cast<float>(std::atof(e));
    if (const char *e = std::getenv("PORESEG_WIDE_BS")) ctx->wide_bs = std::atoi(e) != 0;
    if (const char *e = std::getenv("PORESEG_BRIDGE_SINGLE")) ctx->bridge_single = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("PORESEG_TREE_TAIL")) ctx->tree_tail_pct = std::max(0, std::min(100, std::atoi(e)));
    if (const char *e = std::getenv("PORESEG_FILTER_FUSED")) ctx->filter_fused = std::atoi(e) != 0;
    if (const char *e = std::getenv("PORESEG_UPLOAD")) ctx->upload_by_kernel = std::atoi(e) != 0;
    if (const char *e = std::getenv("PORESEG_TIMING")) ctx->timing = std::max(0, std::min(2, std::atoi(e)));
    if (const char *e = std::getenv("PORESEG_TREE_MW")) ctx->tree_mw = std::atoi(e) != 0;
    if (const char *e = std::getenv("PORESEG_TREE_JPW")) ctx->tree_jobs_per_wave = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("PORESEG_SLOTS_PCT")) ctx->slots_pct = std::max(1, std::min(100, std::atoi(e)));
    if (const char *e = std::getenv("PORESEG_STITCH")) ctx->stitch_host = std::string(e) == "host";
    if (const char *e = std::getenv("PORESEG_TILE")) ctx->tile_len = std::atoll(e);
    if (const char *e = std::getenv("PORESEG_BRIDGE_EXT")) ctx->bridge_ext_on = std::atoi(e) != 0;
    if (const char *e = std::getenv("PORESEG_LAT_HELP")) ctx->lat_help = std::max(0, std::min(2, std::atoi(e)));
    if (const char *e = std::getenv("PORESEG_BRIDGE_BUDGET")) ctx->bridge_budget = std::min(std::max(std::atoi(e), 1), static_cast<int>(BR_MAX));
    if (const char *e = std::getenv("PORESEG_HALO")) ctx->halo = std::atoll(e);
    if (const char *e = std::getenv("PORESEG_DEBUG")) ctx->debug = std::atoi(e) != 0 || e[0] == 0;
    if (const char *e = std::getenv("PORESEG_DBG_K0_NOGRP")) ctx->dbg_k0_nogrp = std::atoi(e) != 0;
}
}  // namespace
#endif

extern "C" {

const char *ps_version(void)
{
#ifdef PS_DIAG
    return "poreseg 0.2 (gfx950) DIAGNOSTIC BUILD";
#else
    return "poreseg 0.2 (gfx950)";
#endif
}

int ps_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ps_create(int device, void *stream, ps_ctx **out)
{
    if (!out) return PS_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return PS_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return PS_ERR_NO_DEVICE;
    ps_ctx *ctx = new ps_ctx();
    ctx->device = device;
    if (stream) {
        ctx->stream = static_cast<hipStream_t>(stream);
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return PS_ERR_HIP; }
        ctx->own_stream = true;
    }
    for (auto &e : ctx->ev)
        if (hipEventCreate(&e) != hipSuccess) { delete ctx; return PS_ERR_HIP; }
    for (auto &e : ctx->ev_front)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { delete ctx; return PS_ERR_HIP; }
    if (ctx->small.reserve(sizeof(SmallLayout) + SMALL_TAIL) != hipSuccess) { delete ctx; return PS_ERR_HIP; }
    ctx->lds_max_samples = (LDS_BYTES_MAX - 1024 * 8 - 256) / (static_cast<int>(sizeof(lds_t)) + 1);   // samples + block sums
    ctx->k0_unaligned = k0_unaligned_probe(device, ctx->stream);
#ifdef PS_DIAG
    diag_env(ctx);                                      // libporeseg_diag.so: the PORESEG_* variables of the experiments (tools/)
#endif
    *out = ctx;
    return PS_OK;
}

void ps_destroy(ps_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf *bufs[] = {&ctx->spine_jobs, &ctx->spine_scratch, &ctx->spine_dense, &ctx->spine_meta, &ctx->tree_jobs,
                      &ctx->tree_scratch, &ctx->tree_spill, &ctx->tree_counts, &ctx->items, &ctx->item_pos,
                      &ctx->first_item, &ctx->ev_off, &ctx->bounds_off, &ctx->small, &ctx->bridges, &ctx->bmeta,
                      &ctx->tile_i32, &ctx->sp_off, &ctx->spine_items, &ctx->asm_hdr, &ctx->ev_first_tile, &ctx->ev_len,
                      &ctx->det_counts, &ctx->det_tics, &ctx->det_cand, &ctx->bsum, &ctx->ev_info, &ctx->chunk_mabs,
                      &ctx->ev_boff, &ctx->blk_mm, &ctx->grp, &ctx->filt_fwd, &ctx->filt_agg, &ctx->filt_zin, &ctx->up_dev,
                      &ctx->align_in, &ctx->align_scratch, &ctx->bridge_ext, &ctx->ext_slot, &ctx->ext_list,
                      &ctx->lat_state, &ctx->lat_seam, &ctx->lat_res, &ctx->pre_c, &ctx->ev_info_tr};
    for (DevBuf *b : bufs) b->release();
    ctx->h_meta.release(); ctx->h_dense.release(); ctx->h_small.release(); ctx->h_up.release(); ctx->h_hdr.release();
    for (auto &e : ctx->ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : ctx->ev_front) if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *ps_last_error(const ps_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int ps_set_tiling(ps_ctx *ctx, int64_t tile_len, int64_t halo)
{
    if (!ctx || tile_len < 0 || halo < 0) return PS_ERR_ARG;
    ctx->tile_len = tile_len;
    ctx->halo = halo;
    return PS_OK;
}

#ifdef PS_DIAG
// (diagnostic library, round 6 experiment) the context's own stream re-created on a subset of the compute units: `count` CUs
// starting at bit `first` of the device's CU mask, every `stride`-th bit, or all the others (hipExtStreamCreateWithCUMask); count 0:
// an ordinary stream again.  What it showed (tools/r6/cu_mask_probe.py, docs/ROUND_6.md): contiguous masks work (64 CUs = two
// XCDs: K0 at 1.9 TB/s), a pool of sixteen masked streams + masked front streams for K0 does not (every masked stream is a
// hardware queue of its own: 0.67 ms per step) -- K0 on CUs of its own is not reachable through stream masks.
static int diag_mask_stream(ps_ctx *ctx, hipStream_t *st, int first, int count, int stride, bool invert)
{
    if (ctx->n_cu <= 0) {
        hipDeviceProp_t prop;
        ctx->n_cu = hipGetDeviceProperties(&prop, ctx->device) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    HIP_TRY(ctx, hipStreamSynchronize(*st));
    HIP_TRY(ctx, hipStreamDestroy(*st));
    *st = nullptr;
    if (count <= 0) { HIP_TRY(ctx, hipStreamCreateWithFlags(st, hipStreamNonBlocking)); return PS_OK; }
    const int words = (ctx->n_cu + 31) / 32;
    std::vector<uint32_t> mask(static_cast<size_t>(words), invert ? 0xffffffffu : 0u);
    for (int k = 0; k < count; ++k) {
        const int b = first + k * std::max(1, stride);
        if (b < 0 || b >= ctx->n_cu) continue;
        if (invert) mask[b >> 5] &= ~(1u << (b & 31)); else mask[b >> 5] |= 1u << (b & 31);
    }
    if (ctx->n_cu & 31) mask[words - 1] &= (1u << (ctx->n_cu & 31)) - 1u;
    HIP_TRY(ctx, hipExtStreamCreateWithCUMask(st, static_cast<uint32_t>(words), mask.data()));
    return PS_OK;
}
#endif

int ps_set_option(ps_ctx *ctx, const char *name, int64_t value)
{
    if (!ctx || !name) return PS_ERR_ARG;
    const std::string n(name);
    if (n == "mode" && value >= 0 && value <= 2) ctx->mode = static_cast<int>(value);
    else if (n == "stitch_host") ctx->stitch_host = value != 0;
    else if (n == "prune") ctx->prune = value != 0;
    else if (n == "scan_bs") ctx->scan_bs = value != 0;
    else if (n == "groups") ctx->groups = value != 0;
    else if (n == "tree_par") ctx->tree_par = value != 0;
    else if (n == "k0_waves" && value >= 0 && value <= 16) ctx->k0_waves = static_cast<int>(value);
    else if (n == "k0_shared") ctx->k0_shared = value != 0;
    else if (n == "k0_admit" && value >= 0) ctx->k0_admit = static_cast<int>(value);
    else if (n == "bridge_ext" && (value == 0 || value == 1)) ctx->bridge_ext_on = static_cast<int>(value);
    // lat_help: 0 every seam walks alone, 1 idle workgroups of the look-ahead kernel help (they leave when nothing has been listed
    // for ~0.1 ms), 2 as 1 but the helpers stay until every workgroup of the launch is through with its own seams -- for a test
    // that must SEE the helpers work; only for a call that has the chip to itself (a helper then waits for nobody)
    else if (n == "lat_help" && value >= 0 && value <= 2) ctx->lat_help = static_cast<int>(value);
    // shared_device: the number of contexts that share this device (a pool of host threads, include/poreseg.h) -- one call that sets
    // what engine.StreamPool set by hand until round 5: n <= 1 a lone context (one-shot K0, helpers on, no admission), n > 1
    // persistent K0 at one wave per SIMD, no helpers, and for n > 3 at most three K0s in flight on the device
    else if (n == "shared_device" && value >= 0 && value <= 4096) {
        const bool sh = value > 1;
        ctx->k0_waves = sh ? 1 : 0;
        ctx->lat_help = sh ? 0 : 1;
        ctx->k0_admit = value > 3 ? 3 : 0;
    }
    else if (n == "single_pass") ctx->single_pass = value != 0;
    else if (n == "gather_fused") ctx->gather_fused = value != 0;
    else if (n == "download_by_kernel") ctx->download_by_kernel = value != 0;
    else if (n == "debug") ctx->debug = value != 0;
    // k0_unaligned 0: K0's fast route from 16-byte-aligned addresses only (what a device whose probe fails gets); 1: whatever the probe said
    else if (n == "k0_unaligned") ctx->k0_unaligned = value != 0 ? k0_unaligned_probe(ctx->device, ctx->stream) : 0;
    else if (n == "slots_pct" && value >= 1 && value <= 100) ctx->slots_pct = static_cast<int>(value);
    else if (n == "tree_jobs_per_wave" && value >= 0 && value <= 1024) ctx->tree_jobs_per_wave = static_cast<int>(value);
    else if (n == "noise_k_ppm" && value >= 0) ctx->noise_k = static_cast<float>(static_cast<double>(value) * 1.0e-6);
#ifdef PS_DIAG
    else if (n == "dbg_phase" && value >= 0 && value <= 2) ctx->dbg_phase = static_cast<int>(value);
    else if (n == "dbg_k0_nogrp") ctx->dbg_k0_nogrp = value != 0;
    // cu_mask: value = first | count << 12 | stride << 24 | invert << 32: the context's OWN stream on those CUs only (count 0: all)
    else if (n == "cu_mask" && ctx->own_stream)
        return diag_mask_stream(ctx, &ctx->stream, static_cast<int>(value & 0xfff), static_cast<int>((value >> 12) & 0xfff),
                                static_cast<int>((value >> 24) & 0xff), ((value >> 32) & 1) != 0);
    else if (n == "k0_sets" && value >= 2 && value <= 4) ctx->k0_sets = static_cast<int>(value);
    else if (n == "scan_lds_pad" && value >= 0 && value <= 65536) ctx->scan_lds_pad = static_cast<int>(value);
    else if (n == "rep_eval" && value >= 1) ctx->rep_eval = static_cast<int>(value);
    else if (n == "rep_stage" && value >= 1) ctx->rep_stage = static_cast<int>(value);
    else if (n == "rep_sum" && value >= 1) ctx->rep_sum = static_cast<int>(value);
#endif
    else if (n == "bridge_budget" && value >= 1 && value <= BR_MAX) ctx->bridge_budget = static_cast<int>(value);
    else if (n == "wide_bs") { ctx->wide_bs = value != 0; ctx->wide_skip = 0; }
    else if (n == "bridge_single" && value >= 1) ctx->bridge_single = static_cast<int>(value);
    else if (n == "tree_tail_pct" && value >= 0 && value <= 100) ctx->tree_tail_pct = static_cast<int>(value);
    else if (n == "filter_fused") ctx->filter_fused = value != 0;
    else if (n == "upload_by_kernel") ctx->upload_by_kernel = value != 0;
    else if (n == "timing") ctx->timing = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(2, value)));
    else if (n == "tree_mw") ctx->tree_mw = value != 0;
    else if (n == "spine_nt" && (value == 256 || value == 512 || value == 1024)) ctx->spine_nt = static_cast<int>(value);
    else if (n == "tree_nt" && (value == 256 || value == 512)) ctx->tree_nt = static_cast<int>(value);
    else return fail(ctx, PS_ERR_ARG, "unknown option or value: %s", name);
    return PS_OK;
}

int ps_synchronize(ps_ctx *ctx)
{
    if (!ctx) return PS_ERR_ARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PS_OK;
}

// cparsers.pyx:55-101
int ps_min_gain(const ps_split_params *p, double *out)
{
    if (!p || !out) return PS_ERR_ARG;
    double fpr = p->false_positive_rate, sps = p->prior_segments_per_second;
    const double sf = p->sampling_freq;
    if (!(fpr != 0.0)) fpr = sf;                                         // :64-65
    if (!(sps != 0.0)) sps = sf / 2.;                                    // :66-67
    if (!(p->max_width >= p->min_width)) return PS_ERR_ASSERT_WIDTH;     // :69
    if (!(p->window_width >= 2 * p->min_width)) return PS_ERR_ASSERT_WINDOW;  // :71
    if (p->cutoff_freq != 0.0 && !(p->cutoff_freq <= 0.5 * sf)) return PS_ERR_ASSERT_CUTOFF;  // :74
    double mg;
    if (p->min_gain_per_sample != 0.0) {
        mg = p->min_gain_per_sample * p->window_width;                   // :84
    } else {
        const double k = p->cutoff_freq != 0.0 ? p->cutoff_freq / (0.5 * sf) : 1.0;   // :89
        mg = (-std::log(sps / (sf - sps)) - std::log(fpr / sf)) / k;     // :95-97
    }
    *out = mg * 2;                                                       // :101
    return PS_OK;
}

int64_t ps_bounds_capacity(const int64_t *h_ev_off, int32_t n_ev, int32_t min_width)
{
    if (!h_ev_off || n_ev < 0 || min_width < 1) return -1;
    int64_t cap = 0;
    for (int e = 0; e < n_ev; ++e) cap += (h_ev_off[e + 1] - h_ev_off[e]) / min_width + 1;
    return cap;
}

int ps_segment_batch(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt,
                     const int64_t *h_ev_off, int32_t n_ev, const ps_split_params *params,
                     int32_t *d_bounds, int64_t cap, int64_t *h_bounds_off, ps_segstat *d_stats)
{
    return ps_segment_batch_ex(ctx, d_samples, fmt, h_ev_off, n_ev, params, d_bounds, cap, h_bounds_off, d_stats, nullptr);
}

int ps_segment_batch_ex(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt,
                        const int64_t *h_ev_off, int32_t n_ev, const ps_split_params *params,
                        int32_t *d_bounds, int64_t cap, int64_t *h_bounds_off, ps_segstat *d_stats,
                        uint8_t *d_is_spine)
{
    if (!ctx) return PS_ERR_ARG;
    if (!h_ev_off || n_ev < 0) return fail(ctx, PS_ERR_ARG, "null/negative argument");
    std::vector<int64_t> st(static_cast<size_t>(n_ev) + 1, 0), ln(static_cast<size_t>(n_ev) + 1, 0);
    for (int e = 0; e < n_ev; ++e) { st[e] = h_ev_off[e]; ln[e] = h_ev_off[e + 1] - h_ev_off[e]; }
    return ps_segment_events(ctx, d_samples, fmt, st.data(), ln.data(), n_ev, params, d_bounds, cap, h_bounds_off,
                             d_stats, d_is_spine);
}

static int segment_events_impl(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt,
                               const int64_t *ev_start, const int64_t *ev_len, int32_t n_ev, const ps_split_params *params,
                               int32_t *d_bounds, int64_t cap, int64_t *h_bounds_off, ps_segstat *d_stats,
                               uint8_t *d_is_spine, bool d_f64);

int ps_segment_events(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt,
                      const int64_t *ev_start, const int64_t *ev_len, int32_t n_ev, const ps_split_params *params,
                      int32_t *d_bounds, int64_t cap, int64_t *h_bounds_off, ps_segstat *d_stats,
                      uint8_t *d_is_spine)
{
    return segment_events_impl(ctx, d_samples, fmt, ev_start, ev_len, n_ev, params, d_bounds, cap, h_bounds_off, d_stats, d_is_spine, false);
}

// The exact route for float64 input on no ADC grid (include/poreseg.h): the reference's own prefix sums -- numpy's sequential
// cumsums, one chain per event (cumsum_ref_kernel) -- and every window scanned with the reference's expressions on them
// (scan_exact_prefix), under the same recursion kernels and device stitch as every other call (LDS-window pipeline: one
// workgroup per window).  Nothing is re-quantised, no sample is read after the cumsums.
int ps_segment_exact_f64(ps_ctx *ctx, const double *d_current, const int64_t *h_ev_start, const int64_t *h_ev_len, int32_t n_ev,
                         const ps_split_params *params, int32_t *d_bounds, int64_t cap, int64_t *h_bounds_off)
{
    if (!ctx) return PS_ERR_ARG;
    const ps_sample_format fmt = {PS_DTYPE_F32, 0, 1.0};        // (unused by the scans of this route: they read c and c2 only)
    return segment_events_impl(ctx, d_current, &fmt, h_ev_start, h_ev_len, n_ev, params, d_bounds, cap, h_bounds_off, nullptr, nullptr, true);
}

static int segment_events_impl(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt,
                               const int64_t *ev_start, const int64_t *ev_len, int32_t n_ev, const ps_split_params *params,
                               int32_t *d_bounds, int64_t cap, int64_t *h_bounds_off, ps_segstat *d_stats,
                               uint8_t *d_is_spine, bool d_f64)
{
    if (!ctx) return PS_ERR_ARG;
    ctx->d_is_spine = d_is_spine;
    const auto t_begin = std::chrono::steady_clock::now();
    if (!ev_start || !ev_len || !params || !h_bounds_off || n_ev < 0 || cap < 0 || (cap > 0 && !d_bounds))
        return fail(ctx, PS_ERR_ARG, "null/negative argument");
    double min_gain = 0;
    int rc = ps_min_gain(params, &min_gain);
    if (rc) return fail(ctx, rc, "reference assertion failed (cparsers.pyx:69-76)");
    const int mw = params->min_width, maxw = params->max_width, W = params->window_width;
    if (mw < 1 || W < 2) return fail(ctx, PS_ERR_ARG, "min_width must be >= 1 and window_width >= 2");
    for (int e = 0; e < n_ev; ++e) {
        const int64_t len = ev_len[e];
        if (len < 0 || ev_start[e] < 0 || len > 0x7fffffff - 2LL * W - 8)
            return fail(ctx, PS_ERR_ARG, "event %d length %lld out of range", e, static_cast<long long>(len));
    }
    {
        int64_t tl = 0;
        for (int e = 0; e < n_ev; ++e) tl += ev_len[e];
        if (tl > 0 && !d_samples) return fail(ctx, PS_ERR_ARG, "d_samples is NULL");
    }
    DevCfg cfg;
    rc = make_cfg(ctx, d_samples, fmt, mw, maxw, W, min_gain, &cfg);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (double &m : ctx->ms) m = 0;
    for (int64_t &c : ctx->counters) c = 0;
    ctx->stream_idle = true;
    if (ctx->k0_shared && hipStreamQuery(ctx->stream) != hipSuccess) { ctx->stream_idle = false; (void)hipGetLastError(); }
    if (ctx->timing >= 1) HIP_TRY(ctx, hipEventRecord(ctx->ev[8], ctx->stream));
    if (ctx->stitch_host) HIP_TRY(ctx, hipMemsetAsync(ctx->small.p, 0, sizeof(SmallLayout), ctx->stream));   // (the device-stitch path clears it with its upload)
    if (d_f64) {
        // exact route: c = cumsum(x), c2 = cumsum(x * x) per event, laid out like the samples (event e at ev_start[e])
        int64_t sample_end = 0;
        for (int e = 0; e < n_ev; ++e) sample_end = std::max(sample_end, ev_start[e] + ev_len[e]);
        const size_t evb = static_cast<size_t>(std::max(1, n_ev)) * sizeof(int64_t);
        HIP_TRY(ctx, ctx->pre_c.reserve(std::max<size_t>(16, 2 * static_cast<size_t>(sample_end) * sizeof(double))));
        HIP_TRY(ctx, ctx->det_cand.reserve(2 * evb));
        HIP_TRY(ctx, ctx->h_hdr.reserve(2 * evb));
        int64_t *h_ev = ctx->h_hdr.as<int64_t>();
        for (int e = 0; e < n_ev; ++e) { h_ev[e] = ev_start[e]; h_ev[n_ev + e] = ev_len[e]; }
        if (n_ev > 0) {
            HIP_TRY(ctx, hipMemcpyAsync(ctx->det_cand.p, h_ev, 2 * static_cast<size_t>(n_ev) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
            double *C = ctx->pre_c.as<double>(), *C2 = C + sample_end;
            hipLaunchKernelGGL(cumsum_ref_kernel, dim3(static_cast<unsigned>(std::min<int64_t>(n_ev, 65535))), dim3(64), 0, ctx->stream,
                               static_cast<const double *>(d_samples), ctx->det_cand.as<int64_t>(), ctx->det_cand.as<int64_t>() + n_ev, n_ev, C, C2);
            HIP_TRY(ctx, hipGetLastError());
            cfg.pre_c = C; cfg.pre_c2 = C2;
        }
    }

    if (!ctx->stitch_host) {
        // block-sum scan: candidates must avoid the ragged ends of a window (min_width >= 8) and a window must fit
        // the single-wave sweep (at most 64 digest chunks: W <= 64 512); otherwise the LDS-window kernels take the call
        bool use_bs = ctx->scan_bs && mw >= 8 && W <= 63 * 1024 && ctx->mode != MODE_EXACT && !d_f64;
        // Counts too wide for the 32-bit digest (K0 says so: RC_WIDE): the call is redone on the 64-bit digest, and if
        // that refuses too (|k - m| >= 2^23) on the LDS-window kernels.  The next 16 calls on the same grid start where
        // this one ended (counters[7]: 1 = 64-bit digest, 2 = LDS-window scan).
        int bs_mode = use_bs ? 1 : 0, redo = 0;
        if (use_bs && ctx->wide_skip > 0 && fmt->quantum == ctx->wide_quantum) { bs_mode = ctx->wide_mode; --ctx->wide_skip; redo = bs_mode == 2 ? 1 : 2; }
        rc = device_stitch_batch(ctx, cfg, bs_mode, ev_start, ev_len, n_ev, mw, W, d_bounds, cap, h_bounds_off, d_stats, t_begin);
        while (rc == RC_WIDE) {
            bs_mode = (bs_mode == 1 && ctx->wide_bs) ? 2 : 0;
            redo = bs_mode == 2 ? 1 : 2;
            ctx->wide_quantum = fmt->quantum;
            ctx->wide_mode = bs_mode;
            ctx->wide_skip = 16;
            for (double &m : ctx->ms) m = 0;
            for (int64_t &c : ctx->counters) c = 0;
            HIP_TRY(ctx, hipMemsetAsync(ctx->small.p, 0, sizeof(SmallLayout), ctx->stream));
            ctx->stream_idle = false;                  // (the front stream, if used, waits for this memset)
            rc = device_stitch_batch(ctx, cfg, bs_mode, ev_start, ev_len, n_ev, mw, W, d_bounds, cap, h_bounds_off, d_stats, t_begin);
        }
        if (redo) ctx->counters[7] = redo;
        if (rc != RC_FALLBACK) return rc;
        // a seam could not be bridged on the device: redo with the host stitch (halo tiles + repairs)
        for (double &m : ctx->ms) m = 0;
        for (int64_t &c : ctx->counters) c = 0;
        ctx->counters[4] = 1000000;
        HIP_TRY(ctx, hipMemsetAsync(ctx->small.p, 0, sizeof(SmallLayout), ctx->stream));
    }

    // ---- tiles ----------------------------------------------------------------------------------
    // Tiling: one spine workgroup per tile.  Default: as many tiles as the chip keeps resident at once
    // (2 workgroups of 512 threads per CU on 256 CUs) so the grid runs as a single wave of blocks,
    // but never shorter than 8 windows (the halo is pure overhead).
    int64_t total_len = 0;
    for (int e = 0; e < n_ev; ++e) total_len += ev_len[e];
    const int64_t H = ctx->halo > 0 ? ctx->halo : 4LL * W;
    int64_t L = ctx->tile_len;
    if (L <= 0) {
        const int64_t resident = 512;
        L = std::max<int64_t>(8LL * W, (total_len + resident - 1) / resident);
    }
    std::vector<SpineJob> jobs;
    std::vector<int64_t> ev_first_tile(static_cast<size_t>(n_ev) + 1, 0);
    int64_t scratch = 0;
    for (int e = 0; e < n_ev; ++e) {
        ev_first_tile[e] = static_cast<int64_t>(jobs.size());
        const int64_t len = ev_len[e];
        if (len == 0) continue;
        const int64_t nt = len <= L + H ? 1 : (len + L - 1) / L;
        for (int64_t t = 0; t < nt; ++t) {
            SpineJob j;
            j.base = ev_start[e];
            j.start = static_cast<int32_t>(t * L);
            j.end = static_cast<int32_t>(len);
            const int64_t stop = (t == nt - 1) ? len : std::min(len, (t + 1) * L + H);
            j.stop = static_cast<int32_t>(stop);
            const int64_t capj = spine_cap(j.start, stop, mw);
            j.out_cap = static_cast<int32_t>(std::min<int64_t>(capj, 0x7fffffff));
            j.out_off = scratch;
            j.first_tile = 0; j.ntiles = 1; j.tile_len = 0x7fffffff; j.ev = e; j.vbase = 0;
            scratch += j.out_cap;
            jobs.push_back(j);
        }
    }
    ev_first_tile[n_ev] = static_cast<int64_t>(jobs.size());
    ctx->counters[2] = static_cast<int64_t>(jobs.size());

    std::vector<TileList> lists;
    rc = run_spines(ctx, cfg, jobs, scratch, lists);
    if (rc) return rc;

    // ---- stitch: true spine per event -------------------------------------------------------------
    const auto t_stitch0 = std::chrono::steady_clock::now();
    std::vector<TreeJob> tjobs;
    std::vector<Item> items;
    std::vector<int64_t> first_item(static_cast<size_t>(n_ev) + 1, 0);
    int64_t tscratch = 0;
    for (int e = 0; e < n_ev; ++e) {
        first_item[e] = static_cast<int64_t>(items.size());
        const int64_t len = ev_len[e];
        const int64_t t0 = ev_first_tile[e], t1 = ev_first_tile[e + 1];
        if (t0 == t1) continue;
        int64_t cur = t0;
        size_t idx = 0;
        int32_t prev = 0;
        for (;;) {
            TileList &Lc = lists[cur];
            if (idx >= Lc.a.size()) {
                if (Lc.ended) break;
                // the chain stopped (passed its stop position) without meeting a later tile:
                // continue it from its last (true) anchor -- "seam repair"
                const int32_t z = Lc.a.empty() ? Lc.start : Lc.a.back().pos;
                std::vector<SpineJob> rj(1);
                rj[0].base = ev_start[e];
                rj[0].start = z;
                rj[0].end = static_cast<int32_t>(len);
                const int64_t stop = std::min<int64_t>(len, static_cast<int64_t>(z) + L + H);
                rj[0].stop = static_cast<int32_t>(stop);
                const int64_t capj = spine_cap(z, stop, mw);
                rj[0].out_cap = static_cast<int32_t>(std::min<int64_t>(capj, 0x7fffffff));
                rj[0].out_off = 0;
                rj[0].first_tile = 0; rj[0].ntiles = 1; rj[0].tile_len = 0x7fffffff; rj[0].ev = e; rj[0].vbase = 0;
                std::vector<TileList> ext;
                rc = run_spines(ctx, cfg, rj, rj[0].out_cap, ext);
                if (rc) return rc;
                ctx->counters[4] += 1;
                Lc.a.insert(Lc.a.end(), ext[0].a.begin(), ext[0].a.end());
                Lc.ended = ext[0].ended;
                if (ext[0].a.empty() && !ext[0].ended)
                    return fail(ctx, PS_ERR_INTERNAL, "seam repair made no progress");
                continue;
            }
            const Anchor an = Lc.a[idx];
            // emit this true anchor
            Item it;
            it.anchor = an.pos;
            it.job = -1;
            if (an.kind == KIND_HIT || an.kind == KIND_LATE) {
                TreeJob tj;
                tj.base = ev_start[e];
                tj.start = prev;
                tj.end = an.pos;
                const int64_t d = static_cast<int64_t>(an.pos) - W - prev;
                tj.j0 = d < 0 ? 0 : static_cast<int32_t>(d / (W / 2)) + 1;
                tj.out_cap = (an.pos - prev) / mw + 1;
                tj.out_off = tscratch;
                tj.m = 0; tj.pad_ = 0; tj.boff = 0;
                tscratch += tj.out_cap;
                it.job = static_cast<int32_t>(tjobs.size());
                tjobs.push_back(tj);
            }
            items.push_back(it);
            prev = an.pos;
            ++idx;
            // does a later tile's speculative spine contain this anchor?  (latest tile wins)
            if (cur + 1 < t1 && an.pos >= lists[cur + 1].start) {
                int64_t u = t0 + std::min<int64_t>(an.pos / L, t1 - t0 - 1);
                for (; u > cur; --u) {
                    const int f = find_in(lists[u], an.pos);
                    if (f != -2) { cur = u; idx = static_cast<size_t>(f + 1); break; }
                }
            }
        }
    }
    first_item[n_ev] = static_cast<int64_t>(items.size());
    const int64_t n_items = static_cast<int64_t>(items.size());
    const size_t n_tj = tjobs.size();
    ctx->counters[3] = static_cast<int64_t>(n_tj);
    const auto t_stitch1 = std::chrono::steady_clock::now();
    ctx->ms[4] = std::chrono::duration<double, std::milli>(t_stitch1 - t_stitch0).count();

    // ---- upload jobs/items, phase 3, gather ------------------------------------------------------
    const size_t up_bytes = n_tj * sizeof(TreeJob) + static_cast<size_t>(n_items) * sizeof(Item) +
                            (static_cast<size_t>(n_ev) + 1) * sizeof(int64_t) * 3;
    HIP_TRY(ctx, ctx->h_up.reserve(up_bytes + 64));
    HIP_TRY(ctx, ctx->tree_jobs.reserve(std::max<size_t>(1, n_tj) * sizeof(TreeJob)));
    // (item_scan_kernel reads counts[item index] speculatively: cover the items as well)
    HIP_TRY(ctx, ctx->tree_counts.reserve(std::max<size_t>(std::max<size_t>(1, n_tj), static_cast<size_t>(n_items)) * sizeof(int32_t)));
    HIP_TRY(ctx, hipMemsetAsync(ctx->tree_counts.p, 0, std::max<size_t>(std::max<size_t>(1, n_tj), static_cast<size_t>(n_items)) * sizeof(int32_t), ctx->stream));   // (tree_job stores only non-zero counts)
    HIP_TRY(ctx, ctx->tree_scratch.reserve(std::max<size_t>(1, static_cast<size_t>(tscratch)) * sizeof(int32_t)));
    HIP_TRY(ctx, ctx->tree_spill.reserve(std::max<size_t>(1, static_cast<size_t>(tscratch)) * sizeof(int2)));
    HIP_TRY(ctx, ctx->items.reserve(std::max<size_t>(1, static_cast<size_t>(n_items)) * sizeof(Item)));
    HIP_TRY(ctx, ctx->item_pos.reserve((static_cast<size_t>(n_items) + 1) * sizeof(int64_t)));
    HIP_TRY(ctx, ctx->first_item.reserve((static_cast<size_t>(n_ev) + 1) * sizeof(int64_t)));
    HIP_TRY(ctx, ctx->ev_off.reserve((static_cast<size_t>(n_ev) + 1) * sizeof(int64_t)));
    HIP_TRY(ctx, ctx->ev_len.reserve((static_cast<size_t>(n_ev) + 1) * sizeof(int64_t)));
    HIP_TRY(ctx, ctx->bounds_off.reserve((static_cast<size_t>(n_ev) + 1) * sizeof(int64_t)));
    char *up = ctx->h_up.as<char>();
    size_t o = 0;
    if (n_tj) {
        std::memcpy(up + o, tjobs.data(), n_tj * sizeof(TreeJob));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->tree_jobs.p, up + o, n_tj * sizeof(TreeJob), hipMemcpyHostToDevice, ctx->stream));
        o += n_tj * sizeof(TreeJob);
    }
    if (n_items) {
        std::memcpy(up + o, items.data(), static_cast<size_t>(n_items) * sizeof(Item));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->items.p, up + o, static_cast<size_t>(n_items) * sizeof(Item), hipMemcpyHostToDevice, ctx->stream));
        o += static_cast<size_t>(n_items) * sizeof(Item);
    }
    o = (o + 7) & ~static_cast<size_t>(7);
    const size_t evb = (static_cast<size_t>(n_ev) + 1) * sizeof(int64_t);
    std::memcpy(up + o, first_item.data(), evb);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->first_item.p, up + o, evb, hipMemcpyHostToDevice, ctx->stream));
    o += evb;
    std::memcpy(up + o, ev_start, evb - sizeof(int64_t));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->ev_off.p, up + o, evb, hipMemcpyHostToDevice, ctx->stream));
    o += evb;
    std::memcpy(up + o, ev_len, evb - sizeof(int64_t));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->ev_len.p, up + o, evb, hipMemcpyHostToDevice, ctx->stream));

    return finish_batch(ctx, cfg, n_tj, n_items, n_ev, d_bounds, cap, h_bounds_off, d_stats, t_begin);
}

static int single_scan(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt, int64_t n, int mode,
                       int mw, double min_gain, double *d_scores, double *gain_out, int32_t *idx_out)
{
    if (!ctx) return PS_ERR_ARG;
    if (!d_samples || n < 0 || n > 0x7fffffff) return fail(ctx, PS_ERR_ARG, "bad samples/n");
    DevCfg cfg;
    int rc = make_cfg(ctx, d_samples, fmt, mw, 0, 0, min_gain, &cfg);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemsetAsync(ctx->small.p, 0, sizeof(SmallLayout), ctx->stream));
    HIP_TRY(ctx, ctx->spine_meta.reserve(64));
    if (d_scores && n) HIP_TRY(ctx, hipMemsetAsync(d_scores, 0, static_cast<size_t>(n) * sizeof(double), ctx->stream));
    SmallLayout *sm = ctx->small.as<SmallLayout>();
    double *d_gain = ctx->spine_meta.as<double>();
    int *d_idx = reinterpret_cast<int *>(d_gain + 1);
    cfg.lds_cap = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(n, ctx->lds_max_samples)));
    {
        const size_t lds = lds_bytes_for(cfg.lds_cap, 1024);
        if (cfg.dtype == PS_DTYPE_F32) {
            HIP_TRY(ctx, set_dyn_lds(ctx, reinterpret_cast<const void *>(single_scan_kernel<1024, PS_DTYPE_F32>), static_cast<int>(lds)));
            hipLaunchKernelGGL((single_scan_kernel<1024, PS_DTYPE_F32>), dim3(1), dim3(1024), lds, ctx->stream, cfg,
                               static_cast<int>(n), mode, d_scores, d_gain, d_idx,
                               reinterpret_cast<unsigned *>(&sm->status), &sm->work0);
        } else {
            HIP_TRY(ctx, set_dyn_lds(ctx, reinterpret_cast<const void *>(single_scan_kernel<1024, PS_DTYPE_I16>), static_cast<int>(lds)));
            hipLaunchKernelGGL((single_scan_kernel<1024, PS_DTYPE_I16>), dim3(1), dim3(1024), lds, ctx->stream, cfg,
                               static_cast<int>(n), mode, d_scores, d_gain, d_idx,
                               reinterpret_cast<unsigned *>(&sm->status), &sm->work0);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, ctx->h_small.reserve(sizeof(SmallLayout)));
    HIP_TRY(ctx, ctx->h_meta.reserve(64));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_meta.p, d_gain, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_small.p, ctx->small.p, sizeof(SmallLayout), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    rc = check_status(ctx, static_cast<unsigned>(ctx->h_small.as<SmallLayout>()->status));
    if (rc) return rc;
    if (gain_out) *gain_out = *ctx->h_meta.as<double>();
    if (idx_out) *idx_out = *reinterpret_cast<int *>(ctx->h_meta.as<double>() + 1);
    return PS_OK;
}

// cparsers.pyx:120-155
int ps_best_single_split(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt, int64_t n,
                         double *gain_out, int32_t *index_out)
{
    if (!gain_out || !index_out) return fail(ctx, PS_ERR_ARG, "null output");
    return single_scan(ctx, d_samples, fmt, n, 1, 1, 0.0, nullptr, gain_out, index_out);
}

// cparsers.pyx:205-249 with no_split=True
int ps_score_window(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt, int64_t n,
                    int32_t min_width, double min_gain, double *d_scores, int32_t *split_out)
{
    if (min_width < 1) return fail(ctx, PS_ERR_ARG, "min_width must be >= 1");
    double g;
    return single_scan(ctx, d_samples, fmt, n, 0, min_width, min_gain, d_scores, &g, split_out);
}


// Diagnostic (tests/test_bound_audit.py): the pruning bounds of the block-sum window scan against the gains they cover,
// on the device, with the product's own scan code (scan_window_bs<.., AUDIT>).  K0 on the one event [0, n), then one wave
// per window.  out[12]: see bs_audit_note (seg_bs.hpp); the three margins are returned as doubles.
int ps_audit_bounds(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt, int64_t n, const ps_split_params *params,
                    const int32_t *h_windows, int32_t n_win, double *out)
{
    if (!ctx) return PS_ERR_ARG;
    if (!d_samples || !params || !h_windows || !out || n < 16 || n > 0x7fffffff || n_win < 1) return fail(ctx, PS_ERR_ARG, "bad argument");
    double mg = 0;
    int rc = ps_min_gain(params, &mg);
    if (rc) return fail(ctx, rc, "invalid split parameters");
    DevCfg cfg;
    rc = make_cfg(ctx, d_samples, fmt, params->min_width, params->max_width, params->window_width, mg, &cfg);
    if (rc) return rc;
    if (params->min_width < 8) return fail(ctx, PS_ERR_ARG, "the block-sum scan needs min_width >= 8");
    for (int i = 0; i < n_win; ++i)
        if (h_windows[2 * i] < 0 || h_windows[2 * i + 1] > n || h_windows[2 * i + 1] - h_windows[2 * i] <= 2 * params->min_width)
            return fail(ctx, PS_ERR_ARG, "window %d: [%d, %d) is not a window the recursion would scan", i, h_windows[2 * i], h_windows[2 * i + 1]);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    cfg.mode = MODE_FAST;
    const int64_t nb_total = (n + 7) / 8, nb_pad = k0_padded_blocks(nb_total);
    const int64_t tab[4] = {0, n, 0, nb_total};                            // ev_start | ev_len | ev_boff[0..1]
    const size_t win_bytes = static_cast<size_t>(n_win) * 2 * sizeof(int32_t);
    HIP_TRY(ctx, ctx->bsum.reserve(std::max<size_t>(16384, static_cast<size_t>(nb_pad) * sizeof(uint2))));
    HIP_TRY(ctx, ctx->ev_info.reserve(sizeof(int4)));
    HIP_TRY(ctx, ctx->chunk_mabs.reserve(static_cast<size_t>(nb_pad / BS_CHUNK + 1) * sizeof(int4)));
    HIP_TRY(ctx, ctx->grp.reserve(static_cast<size_t>(nb_pad / BS_GRP + 1) * sizeof(uint4)));
    HIP_TRY(ctx, ctx->up_dev.reserve(sizeof(tab) + 128 + win_bytes));
    HIP_TRY(ctx, ctx->h_small.reserve(sizeof(SmallLayout)));
    char *d = ctx->up_dev.as<char>();
    unsigned long long acc[12] = {0, 0, 0, 0, 0, 0, 0x7fffffffull, 0x7fffffffull, 0x7fffffffull, 0, 0, 0};
    HIP_TRY(ctx, hipMemsetAsync(ctx->small.p, 0, sizeof(SmallLayout), ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d, tab, sizeof(tab), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d + 32, acc, sizeof(acc), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d + 128, h_windows, win_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));                        // (the host arrays are the caller's / the stack's)
    SmallLayout *sm = ctx->small.as<SmallLayout>();
    const int64_t *dt = reinterpret_cast<const int64_t *>(d);
    cfg.grp = ctx->grp.p;
    const unsigned k0_grid = static_cast<unsigned>((nb_pad / K0_WB + K0_WAVES - 1) / K0_WAVES);
    const bool f32 = cfg.dtype == PS_DTYPE_F32;
#define PS_K0A(DTV) hipLaunchKernelGGL((blocksum_kernel<DTV>), dim3(k0_grid), dim3(64 * K0_WAVES), 0, ctx->stream, cfg, dt, dt + 1, dt + 2, 1, n, \
                                     ctx->bsum.p, ctx->ev_info.as<int4>(), ctx->chunk_mabs.as<int4>(), reinterpret_cast<unsigned *>(&sm->status), \
                                     static_cast<uint4 *>(ctx->grp.p))
    if (f32) PS_K0A(PS_DTYPE_F32); else PS_K0A(PS_DTYPE_I16);
#undef PS_K0A
    HIP_TRY(ctx, hipGetLastError());
    cfg.bsum = ctx->bsum.p;
    cfg.ev_info = ctx->ev_info.as<int4>();
    cfg.chunk_tot = ctx->chunk_mabs.as<int4>();
    cfg.dbg = reinterpret_cast<unsigned long long *>(d + 32);
    const unsigned g = static_cast<unsigned>(std::min<int32_t>(n_win, 4096));
    if (f32) hipLaunchKernelGGL((audit_kernel<PS_DTYPE_F32>), dim3(g), dim3(64), 0, ctx->stream, cfg, reinterpret_cast<const int2 *>(d + 128), n_win,
                                reinterpret_cast<unsigned *>(&sm->status));
    else     hipLaunchKernelGGL((audit_kernel<PS_DTYPE_I16>), dim3(g), dim3(64), 0, ctx->stream, cfg, reinterpret_cast<const int2 *>(d + 128), n_win,
                                reinterpret_cast<unsigned *>(&sm->status));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_small.p, ctx->small.p, sizeof(SmallLayout), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(acc, d + 32, sizeof(acc), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->tile_cache.valid = false;                                          // (the tables of the last batch call are gone)
    const unsigned st = static_cast<unsigned>(ctx->h_small.as<SmallLayout>()->status);
    if (st & ST_WIDE_RANGE) return fail(ctx, PS_ERR_ARG, "counts too wide for the 32-bit digest: nothing to audit");
    rc = check_status(ctx, st);
    if (rc) return rc;
    for (int i = 0; i < 12; ++i) out[i] = static_cast<double>(acc[i]);
    for (int i = 6; i < 9; ++i) {
        int k = static_cast<int>(static_cast<unsigned>(acc[i] & 0xffffffffull));
        if (k == 0x7fffffff) { out[i] = INFINITY; continue; }
        k ^= (k >> 31) & 0x7fffffff;
        float f;
        std::memcpy(&f, &k, sizeof(f));
        out[i] = static_cast<double>(f);
    }
    return PS_OK;
}

// Replaces lambda_event_parser.parse with the default rules (parsers.py:124-155).
namespace {
// The detector after its streaming pass: edges (sample positions where the mask flips) -> pieces longer than min_duration
// (parsers.py:133) -> their extremes (piece_minmax_kernel: whole 4 096-sample chunks from the table the streaming pass left in
// det_counts, the ragged ends from the samples) -> the rules min > min_current, max < threshold (:134-135).
int events_from_edges(ps_ctx *ctx, const DevCfg &cfg, int64_t n, std::vector<int> &tics, double threshold, int64_t min_duration,
                      double min_current, int64_t *h_starts, int64_t *h_lengths, int64_t cap, int64_t *n_events_out)
{
    const bool f32 = cfg.dtype == PS_DTYPE_F32;
    std::sort(tics.begin(), tics.end());
    std::vector<int2> cand;
    int a = 0;
    for (size_t p = 0; p <= tics.size(); ++p) {
        const int b = p == tics.size() ? static_cast<int>(n) : tics[p];
        if (static_cast<int64_t>(b) - a > min_duration) cand.push_back(make_int2(a, b));
        a = b;
    }
    const size_t nc = cand.size();
    if (nc == 0) return PS_OK;
    HIP_TRY(ctx, ctx->det_cand.reserve(nc * sizeof(int2) * 2));
    HIP_TRY(ctx, ctx->h_up.reserve(nc * sizeof(int2)));
    std::memcpy(ctx->h_up.p, cand.data(), nc * sizeof(int2));
    int2 *d_cand = ctx->det_cand.as<int2>(), *d_mm = d_cand + nc;
    HIP_TRY(ctx, hipMemcpyAsync(d_cand, ctx->h_up.p, nc * sizeof(int2), hipMemcpyHostToDevice, ctx->stream));
    if (f32) hipLaunchKernelGGL((piece_minmax_kernel<PS_DTYPE_F32>), dim3(static_cast<unsigned>(nc)), dim3(256), 0, ctx->stream, cfg, d_cand, static_cast<int>(nc), ctx->det_counts.as<int2>(), d_mm);
    else     hipLaunchKernelGGL((piece_minmax_kernel<PS_DTYPE_I16>), dim3(static_cast<unsigned>(nc)), dim3(256), 0, ctx->stream, cfg, d_cand, static_cast<int>(nc), ctx->det_counts.as<int2>(), d_mm);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, ctx->h_dense.reserve(nc * sizeof(int2)));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_dense.p, d_mm, nc * sizeof(int2), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int2 *hm = ctx->h_dense.as<int2>();
    int64_t kept = 0;
    for (size_t i = 0; i < nc; ++i) {
        const double mn = static_cast<double>(hm[i].x) * cfg.q, mx = static_cast<double>(hm[i].y) * cfg.q;
        if (mn > min_current && mx < threshold) {                                        // parsers.py:134-135
            if (kept < cap) { h_starts[kept] = cand[i].x; h_lengths[kept] = cand[i].y - cand[i].x; }
            ++kept;
        }
    }
    *n_events_out = kept;
    if (kept > cap) return fail(ctx, PS_ERR_CAPACITY, "event capacity %lld < %lld", static_cast<long long>(cap), static_cast<long long>(kept));
    return PS_OK;
}

// One streaming pass that leaves the edge list in `tics` and min / max per DET_CHUNK samples in det_counts.  by_blocks = false:
// edge_scan_kernel over the samples; true (ps_detect_segment_trace): edge_cls_kernel over the per-block verdicts K0 left.
int detect_edges(ps_ctx *ctx, const DevCfg &cfg, int64_t n, double threshold, bool by_blocks, std::vector<int> &tics,
                 unsigned *status_out)
{
    const int nb = static_cast<int>((n + DET_CHUNK - 1) / DET_CHUNK);
    const bool f32 = cfg.dtype == PS_DTYPE_F32;
    SmallLayout *sm = ctx->small.as<SmallLayout>();
    unsigned *d_ntics = reinterpret_cast<unsigned *>(&sm->dense);
    HIP_TRY(ctx, ctx->det_counts.reserve(static_cast<size_t>(nb) * sizeof(int2)));
    HIP_TRY(ctx, ctx->h_small.reserve(sizeof(SmallLayout)));
    size_t tics_cap = std::max<size_t>(ctx->det_tics.cap / sizeof(int), 1u << 16);
    for (int attempt = 0; attempt < 2; ++attempt) {
        HIP_TRY(ctx, ctx->det_tics.reserve(tics_cap * sizeof(int)));
        // (by_blocks: K0 ran just before on this stream and its status bits -- off grid, counts too wide -- are still wanted; the
        //  edge count is still zero from the call's setup kernel on the first attempt)
        if (!by_blocks) HIP_TRY(ctx, hipMemsetAsync(ctx->small.p, 0, sizeof(SmallLayout), ctx->stream));
        else if (attempt > 0) HIP_TRY(ctx, hipMemsetAsync(&sm->dense, 0, sizeof(unsigned long long), ctx->stream));
        const unsigned cls_grid = static_cast<unsigned>((((n + 7) / 8 + 63) / 64 + CLS_NT - 1) / CLS_NT);
#define PS_EDGE(DTV)                                                                                                                        \
        do {                                                                                                                                \
            if (by_blocks) hipLaunchKernelGGL((edge_cls_kernel<DTV>), dim3(cls_grid), dim3(CLS_NT), 0, ctx->stream, cfg, n, threshold,      \
                                              ctx->blk_cls.as<unsigned char>(), ctx->cls_mm.as<int2>(), ctx->ev_info.as<int4>(),            \
                                              ctx->det_tics.as<int>(), d_ntics, static_cast<unsigned>(tics_cap),                            \
                                              ctx->det_counts.as<int2>(), reinterpret_cast<unsigned *>(&sm->status));                       \
            else hipLaunchKernelGGL((edge_scan_kernel<DTV>), dim3(nb), dim3(DET_NT), 0, ctx->stream, cfg, n, threshold,                     \
                                    ctx->det_tics.as<int>(), d_ntics, static_cast<unsigned>(tics_cap), ctx->det_counts.as<int2>(),          \
                                    reinterpret_cast<unsigned *>(&sm->status));                                                             \
        } while (0)
        if (f32) PS_EDGE(PS_DTYPE_F32); else PS_EDGE(PS_DTYPE_I16);
#undef PS_EDGE
        HIP_TRY(ctx, hipGetLastError());
        // speculative copy of the first edges together with the count: one sync in the common case
        const size_t spec = std::min<size_t>(tics_cap, 4096);
        HIP_TRY(ctx, ctx->h_dense.reserve(std::max(spec, static_cast<size_t>(1)) * sizeof(int)));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_small.p, ctx->small.p, sizeof(SmallLayout), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_dense.p, ctx->det_tics.p, spec * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const SmallLayout hs = *ctx->h_small.as<SmallLayout>();
        if (status_out) *status_out = static_cast<unsigned>(hs.status);
        if (by_blocks && (static_cast<unsigned>(hs.status) & ST_WIDE_RANGE)) return PS_OK;      // (the caller takes the other route)
        const int rc = check_status(ctx, static_cast<unsigned>(hs.status));
        if (rc) return rc;
        const size_t ne = static_cast<unsigned>(hs.dense);
        if (ne > tics_cap) { tics_cap = ne + 1024; continue; }          // list overflowed: rerun with room for all edges
        tics.resize(ne);
        if (ne <= spec) std::memcpy(tics.data(), ctx->h_dense.p, ne * sizeof(int));
        else HIP_TRY(ctx, hipMemcpy(tics.data(), ctx->det_tics.p, ne * sizeof(int), hipMemcpyDeviceToHost));
        break;
    }
    return PS_OK;
}
}  // namespace

int ps_detect_events(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt, int64_t n,
                     double threshold, int64_t min_duration, double min_current,
                     int64_t *h_starts, int64_t *h_lengths, int64_t cap, int64_t *n_events_out)
{
    if (!ctx) return PS_ERR_ARG;
    if (!n_events_out || n < 0 || n > 0x7fffffff || cap < 0 || (cap > 0 && (!h_starts || !h_lengths)))
        return fail(ctx, PS_ERR_ARG, "bad argument");
    *n_events_out = 0;
    if (n == 0) return PS_OK;
    if (!d_samples) return fail(ctx, PS_ERR_ARG, "d_samples is NULL");
    DevCfg cfg;
    int rc = make_cfg(ctx, d_samples, fmt, 1, 1, 2, 0.0, &cfg);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<int> tics;
    rc = detect_edges(ctx, cfg, n, threshold, false, tics, nullptr);
    if (rc) return rc;
    return events_from_edges(ctx, cfg, n, tics, threshold, min_duration, min_current, h_starts, h_lengths, cap, n_events_out);
}

// File.parse + Event.parse for a whole file trace with ONE pass over its samples (round 6).  The two calls
// ps_detect_events + ps_segment_events stream the samples twice: once for the detector's edges and extremes, once -- the 92 %
// of them that lie in events -- for K0's block sums.  Here K0 runs over the WHOLE trace as one event (its blocks aligned to the
// trace) and judges every block against the detector's threshold on its way (DevCfg::blk_cls: 2 bits per block, min / max per 128
// blocks), the detector reads those bits -- 1/64 of the samples' bytes -- instead of the samples (edge_cls_kernel), and the
// events it cuts out -- they start at any sample -- are segmented from the same digest: an event that starts ph = start mod 8
// samples into a block is scanned in coordinates shifted by ph (EvRef::ph, scan_window_ph).  Same events, same boundaries, same
// statistics (to the last bits of the final fp64 expressions) as the two calls (tests/test_single_pass.py, tools/r6/fuzz_single_pass.py).  Falls back to the two calls by itself
// when the block-sum scan does not apply (min_width < 8, W > 64 512, options) or the trace's counts leave the 32-bit digest
// about its first sample (|k - k_0| >= 2^14: the sums are centred on the trace's first sample here, not on each event's).
int ps_detect_segment_trace(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt, int64_t n,
                            double threshold, int64_t min_duration, double min_current, const ps_split_params *params,
                            int64_t *h_starts, int64_t *h_lengths, int64_t ev_cap, int64_t *n_events_out,
                            int32_t *d_bounds, int64_t cap, int64_t *h_bounds_off, ps_segstat *d_stats)
{
    if (!ctx) return PS_ERR_ARG;
    if (!n_events_out || !params || !h_bounds_off || n < 0 || n > 0x7fffffff - 16 || ev_cap < 0 || cap < 0 ||
        (ev_cap > 0 && (!h_starts || !h_lengths)) || (cap > 0 && !d_bounds))
        return fail(ctx, PS_ERR_ARG, "bad argument");
    *n_events_out = 0;
    h_bounds_off[0] = 0;
    if (n == 0) return PS_OK;
    if (!d_samples) return fail(ctx, PS_ERR_ARG, "d_samples is NULL");
    const auto t_begin = std::chrono::steady_clock::now();
    double min_gain = 0;
    int rc = ps_min_gain(params, &min_gain);
    if (rc) return fail(ctx, rc, "reference assertion failed (cparsers.pyx:69-76)");
    const int mw = params->min_width, maxw = params->max_width, W = params->window_width;
    if (mw < 1 || W < 2) return fail(ctx, PS_ERR_ARG, "min_width must be >= 1 and window_width >= 2");
    DevCfg cfg;
    rc = make_cfg(ctx, d_samples, fmt, mw, maxw, W, min_gain, &cfg);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    auto two_calls = [&]() -> int {
        int r2 = ps_detect_events(ctx, d_samples, fmt, n, threshold, min_duration, min_current, h_starts, h_lengths, ev_cap, n_events_out);
        if (r2) return r2;
        return ps_segment_events(ctx, d_samples, fmt, h_starts, h_lengths, static_cast<int32_t>(*n_events_out), params, d_bounds, cap,
                                 h_bounds_off, d_stats, nullptr);
    };
    const bool use_bs = ctx->scan_bs && mw >= 8 && W <= 63 * 1024 && ctx->mode != MODE_EXACT && !ctx->stitch_host && ctx->single_pass &&
                        !(ctx->wide_skip > 0 && fmt->quantum == ctx->wide_quantum);
    if (!use_bs) return two_calls();
    for (double &m : ctx->ms) m = 0;
    for (int64_t &c : ctx->counters) c = 0;
    ctx->d_is_spine = nullptr;
    if (ctx->timing >= 1) HIP_TRY(ctx, hipEventRecord(ctx->ev[8], ctx->stream));
    // ---- K0 over the whole trace: one event [0, n), blocks aligned to the trace, per-block extremes on -------------------------
    const bool f32 = cfg.dtype == PS_DTYPE_F32;
    SmallLayout *sm = ctx->small.as<SmallLayout>();
    const int64_t nb_total = (n + 7) / 8, nb_pad = k0_padded_blocks(nb_total);
    HIP_TRY(ctx, ctx->bsum.reserve(std::max<size_t>(16384, static_cast<size_t>(nb_pad) * sizeof(uint2))));
    HIP_TRY(ctx, ctx->ev_info.reserve(sizeof(int4)));
    HIP_TRY(ctx, ctx->chunk_mabs.reserve(static_cast<size_t>(nb_pad / BS_CHUNK + 1) * sizeof(int4)));
    if (d_stats) HIP_TRY(ctx, ctx->blk_mm.reserve(static_cast<size_t>(nb_pad) * sizeof(int)));
    if (ctx->groups) HIP_TRY(ctx, ctx->grp.reserve(static_cast<size_t>(nb_pad / BS_GRP + 1) * sizeof(uint4)));
    HIP_TRY(ctx, ctx->blk_cls.reserve(static_cast<size_t>(nb_pad / 4 + 64)));
    HIP_TRY(ctx, ctx->cls_mm.reserve(static_cast<size_t>(nb_pad / BS_CHUNK + 2) * sizeof(int2)));
    HIP_TRY(ctx, ctx->det_cand.reserve(4 * sizeof(int64_t)));
    // below(k) <=> double(k) * q < threshold (below_thr): monotone in k, so there is ONE integer kthr with below(k) <=> k < kthr
    int kthr;
    {
        const double q = cfg.q;
        double kk = std::ceil(threshold / q);
        kk = std::max(-2147483000.0, std::min(2147483000.0, kk));
        long long k = static_cast<long long>(kk);
        while (k > -2147483000LL && !(static_cast<double>(k - 1) * q < threshold)) --k;
        while (k < 2147483000LL && static_cast<double>(k) * q < threshold) ++k;
        kthr = static_cast<int>(k);
    }
    // (one launch: the status block cleared, the table of the call's one event -- the whole trace -- written)
    hipLaunchKernelGGL(trace_setup_kernel, dim3(1), dim3(256), 0, ctx->stream, reinterpret_cast<unsigned long long *>(ctx->small.p),
                       static_cast<int>(sizeof(SmallLayout) / sizeof(unsigned long long)), ctx->det_cand.as<long long>(),
                       static_cast<long long>(n), static_cast<long long>(nb_total));
    HIP_TRY(ctx, hipGetLastError());
    cfg.grp = ctx->groups ? ctx->grp.p : nullptr;
    cfg.blk_mm = d_stats ? ctx->blk_mm.as<int>() : nullptr;
    cfg.blk_cls = ctx->blk_cls.as<unsigned char>();
    cfg.cls_mm = ctx->cls_mm.as<int2>();
    cfg.cls_kthr = kthr;
    if (ctx->n_cu <= 0) {
        hipDeviceProp_t prop;
        ctx->n_cu = hipGetDeviceProperties(&prop, ctx->device) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    {
        const unsigned k0_full = static_cast<unsigned>((nb_pad / K0_WB + K0_WAVES - 1) / K0_WAVES);
        const unsigned k0_per_cu = static_cast<unsigned>(ctx->k0_waves) * (f32 ? 1u : 2u);
        const unsigned k0_grid = ctx->k0_waves > 0 ? std::min(k0_full, k0_per_cu * static_cast<unsigned>(ctx->n_cu) * (4u / K0_WAVES)) : k0_full;
        const int64_t *d_ev = ctx->det_cand.as<int64_t>();
        if (ctx->k0_admit > 0) {
            const int grc = chain_enter(ctx, ctx->device, ctx->k0_admit, ctx->stream, &ctx->chain_ticket);
            if (grc) return grc;
        }
#define PS_K0T(DTV) hipLaunchKernelGGL((blocksum_kernel<DTV, 2>), dim3(k0_grid), dim3(64 * K0_WAVES), 0, ctx->stream, cfg, d_ev, d_ev + 1, d_ev + 2, 1, n, \
                                     ctx->bsum.p, ctx->ev_info.as<int4>(), ctx->chunk_mabs.as<int4>(), reinterpret_cast<unsigned *>(&sm->status),          \
                                     const_cast<uint4 *>(static_cast<const uint4 *>(cfg.grp)))
        if (f32) PS_K0T(PS_DTYPE_F32); else PS_K0T(PS_DTYPE_I16);
#undef PS_K0T
        HIP_TRY(ctx, hipGetLastError());
        if (ctx->k0_admit > 0) { const int crc = chain_publish(ctx, ctx->device, ctx->stream, ctx->chain_ticket); if (crc) return crc; }
    }
    // ---- the detector on K0's verdicts per block ----------------------------------------------------------------------------
    std::vector<int> tics;
    unsigned st = 0;
    rc = detect_edges(ctx, cfg, n, threshold, true, tics, &st);
    if (rc) return rc;
    if (st & ST_WIDE_RANGE) { ctx->counters[7] = 3; const int r2 = two_calls(); ctx->counters[7] = 3; return r2; }    // counts too wide about the trace's first sample
    rc = events_from_edges(ctx, cfg, n, tics, threshold, min_duration, min_current, h_starts, h_lengths, ev_cap, n_events_out);
    if (rc) return rc;
    const int32_t n_ev = static_cast<int32_t>(*n_events_out);
    if (n_ev == 0) return PS_OK;
    for (int e = 0; e < n_ev; ++e)
        if (h_lengths[e] > 0x7fffffff - 2LL * W - 16) return fail(ctx, PS_ERR_ARG, "event %d length %lld out of range", e, static_cast<long long>(h_lengths[e]));
    // ---- every event from the trace's digest ----------------------------------------------------------------------------------
    cfg.bsum = ctx->bsum.p;
    cfg.blk_cls = nullptr;
    rc = device_stitch_batch(ctx, cfg, 1, h_starts, h_lengths, n_ev, mw, W, d_bounds, cap, h_bounds_off, d_stats, t_begin, true);
    if (rc == RC_FALLBACK || rc == RC_WIDE) {          // (a seam the device could not mend: the two calls, with the host stitch behind them)
        return ps_segment_events(ctx, d_samples, fmt, h_starts, h_lengths, n_ev, params, d_bounds, cap, h_bounds_off, d_stats, nullptr);
    }
    return rc;
}

int ps_get_timings(const ps_ctx *ctx, double *ms, int32_t n_ms, int64_t *counters, int32_t n_counters)
{
    if (!ctx) return PS_ERR_ARG;
    for (int i = 0; i < n_ms && ms; ++i) ms[i] = i < 8 ? ctx->ms[i] : 0.0;
    for (int i = 0; i < n_counters && counters; ++i) counters[i] = i < 14 ? ctx->counters[i] : 0;
    return PS_OK;
}

// The context's fourteen work counters where ps_get_timings copies them from: a host that looks at one of them after every
// call (the near-tie count, counters[11]) reads it in place instead of making a second call.
const int64_t *ps_counters(const ps_ctx *ctx) { return ctx ? ctx->counters : nullptr; }

int ps_synth_trace(ps_ctx *ctx, void *d_out, int32_t dtype, int64_t n, uint64_t seed,
                   const int64_t *h_seg_end, const int32_t *h_level_counts, int64_t nseg)
{
    if (!ctx) return PS_ERR_ARG;
    if (!d_out || n < 0 || nseg < 1 || !h_seg_end || !h_level_counts) return fail(ctx, PS_ERR_ARG, "bad argument");
    if (h_seg_end[nseg - 1] < n) return fail(ctx, PS_ERR_ARG, "segment table does not cover n");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t b1 = static_cast<size_t>(nseg) * sizeof(int64_t), b2 = static_cast<size_t>(nseg) * sizeof(int32_t);
    HIP_TRY(ctx, ctx->tree_scratch.reserve(b1 + b2));
    char *d = ctx->tree_scratch.as<char>();
    HIP_TRY(ctx, hipMemcpyAsync(d, h_seg_end, b1, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d + b1, h_level_counts, b2, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // pageable host sources must outlive the copy
    hipLaunchKernelGGL(synth_kernel, dim3(4096), dim3(256), 0, ctx->stream, d_out, dtype, n,
                       static_cast<unsigned long long>(seed), reinterpret_cast<const int64_t *>(d),
                       reinterpret_cast<const int32_t *>(d + b1), nseg);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PS_OK;
}

namespace {
// scipy.signal.bessel(N, wn, 'low', analog=False, output='ba') for N = 2..8: poles of the phase-normalised analog
// prototype (what scipy.signal.besselap(N, 'phase') returns; conjugates implied), lp2lp with the pre-warped frequency,
// bilinear transform with fs = 2, polynomial expansion.  Then lfilter_zi (steady state of the delays for a unit step:
// (I - A) zi = B), the halo from the powers of the state matrix, and one launch of filt_halo_kernel.
int filter_order_n(ps_ctx *ctx, const DevCfg &cfg, int64_t n, int order, double wn, double *d_out)
{
    typedef std::complex<double> cd;
    static const double poles[FILT_MAXORD + 1][4][2] = {       // [order][pair][re, im]; im == 0: a single real pole
        {{0, 0}}, {{-0.9999999999999998, 0.0}},
        {{-0.8660254037844384, 0.4999999999999999}},
        {{-0.9416000265332067, 0.0}, {-0.7456403858480766, 0.7113666249728351}},
        {{-0.9047587967882447, 0.27091873300387465}, {-0.6572111716718827, 0.830161435004873}},
        {{-0.9264420773877602, 0.0}, {-0.8515536193688396, 0.44271746394433265}, {-0.5905759446119191, 0.9072067564574549}},
        {{-0.9093906830472273, 0.1856964396793047}, {-0.7996541858328288, 0.5621717346937318}, {-0.5385526816693109, 0.9616876881954278}},
        {{-0.919487155649029, 0.0}, {-0.8800029341523375, 0.32166527623077396}, {-0.7527355434093214, 0.6504696305522552},
         {-0.4966917256672317, 1.0025085084544205}},
        {{-0.909683154665291, 0.1412437976671423}, {-0.8473250802359334, 0.42590175382729345}, {-0.7111381808485397, 0.7186517314108402},
         {-0.4621740412532123, 1.0343886811269012}}};
    const double wo = 4.0 * std::tan(3.14159265358979323846 * wn / 2.0);
    std::vector<cd> p;
    for (int k = 0; k < (order + 1) / 2; ++k) {
        const double re = poles[order][k][0], im = poles[order][k][1];
        if (im == 0.0) p.push_back(cd(re * wo, 0.0));
        else { p.push_back(cd(re, im) * wo); p.push_back(cd(re, -im) * wo); }
    }
    cd den(1.0, 0.0);
    for (const cd &q : p) den *= (cd(4.0, 0.0) - q);
    const double gain = std::pow(wo, order) * (cd(1.0, 0.0) / den).real();
    std::vector<cd> ac(order + 1, cd(0.0, 0.0));
    ac[0] = cd(1.0, 0.0);
    for (int k = 0; k < order; ++k) {
        const cd pz = (cd(4.0, 0.0) + p[k]) / (cd(4.0, 0.0) - p[k]);
        for (int j = k + 1; j >= 1; --j) ac[j] = ac[j] - pz * ac[j - 1];
    }
    std::vector<double> bc(order + 1, 0.0);
    bc[0] = 1.0;
    for (int k = 0; k < order; ++k)
        for (int j = k + 1; j >= 1; --j) bc[j] += bc[j - 1];
    FiltN f = {};
    f.order = order; f.pad = 3 * (order + 1);
    for (int j = 0; j <= order; ++j) { f.a[j] = ac[j].real(); f.b[j] = gain * bc[j]; }
    // state matrix A (z' = A z + B x) and zi = (I - A)^-1 B by Gaussian elimination with pivoting
    double A[FILT_MAXORD][FILT_MAXORD] = {}, M[FILT_MAXORD][FILT_MAXORD + 1] = {};
    for (int r = 0; r < order; ++r) {
        A[r][0] = -f.a[r + 1];
        if (r + 1 < order) A[r][r + 1] += 1.0;
        for (int c2 = 0; c2 < order; ++c2) M[r][c2] = (r == c2 ? 1.0 : 0.0) - A[r][c2];
        M[r][order] = f.b[r + 1] - f.a[r + 1] * f.b[0];
    }
    for (int c2 = 0; c2 < order; ++c2) {
        int piv = c2;
        for (int r = c2 + 1; r < order; ++r) if (std::fabs(M[r][c2]) > std::fabs(M[piv][c2])) piv = r;
        if (M[piv][c2] == 0.0) return fail(ctx, PS_ERR_ARG, "filter has no steady state");
        for (int k = 0; k <= order; ++k) std::swap(M[c2][k], M[piv][k]);
        for (int r = 0; r < order; ++r) {
            if (r == c2) continue;
            const double g = M[r][c2] / M[c2][c2];
            for (int k = c2; k <= order; ++k) M[r][k] -= g * M[c2][k];
        }
    }
    for (int r = 0; r < order; ++r) f.zi[r] = M[r][order] / M[r][r];
    // halo: smallest multiple of 64 with ||A^H||_inf <= 2^-70 (A^64 by squaring, then one factor per step)
    auto mul = [&](const double X[FILT_MAXORD][FILT_MAXORD], const double Y[FILT_MAXORD][FILT_MAXORD], double Z[FILT_MAXORD][FILT_MAXORD]) {
        double T[FILT_MAXORD][FILT_MAXORD] = {};
        for (int i = 0; i < order; ++i) for (int j = 0; j < order; ++j) for (int k = 0; k < order; ++k) T[i][j] += X[i][k] * Y[k][j];
        std::memcpy(Z, T, sizeof(T));
    };
    double A64[FILT_MAXORD][FILT_MAXORD], P[FILT_MAXORD][FILT_MAXORD];
    std::memcpy(A64, A, sizeof(A));
    for (int k = 0; k < 6; ++k) mul(A64, A64, A64);
    std::memcpy(P, A64, sizeof(P));
    int H = 0;
    for (int h = 64; h <= 8192; h += 64) {
        double nrm = 0.0;
        for (int i = 0; i < order; ++i) { double r = 0.0; for (int j = 0; j < order; ++j) r += std::fabs(P[i][j]); nrm = std::max(nrm, r); }
        if (nrm <= 8.47e-22) { H = h; break; }          // 2^-70
        mul(P, A64, P);
    }
    if (!H) return fail(ctx, PS_ERR_ARG, "a Bessel filter of order %d this slow (cutoff / Nyquist = %g) does not run on the device", order, wn);
    const int S = std::max(1024, 4 * H);
    const int64_t m = n + 2LL * f.pad, nseg = (m + S - 1) / S;
    HIP_TRY(ctx, ctx->filt_fwd.reserve(static_cast<size_t>(nseg) * static_cast<size_t>(S + H) * sizeof(double)));
    HIP_TRY(ctx, ctx->h_small.reserve(sizeof(SmallLayout)));
    if (!ctx->defer_sync) HIP_TRY(ctx, hipMemsetAsync(ctx->small.p, 0, sizeof(SmallLayout), ctx->stream));
    unsigned *st = reinterpret_cast<unsigned *>(&ctx->small.as<SmallLayout>()->status);
    const dim3 grid(static_cast<unsigned>((nseg + 63) / 64));
    if (cfg.dtype == PS_DTYPE_F32)      hipLaunchKernelGGL((filt_halo_kernel<PS_DTYPE_F32>), grid, dim3(64), 0, ctx->stream, cfg, f, n, S, H, ctx->filt_fwd.as<double>(), d_out, st);
    else if (cfg.dtype == PS_DTYPE_F64) hipLaunchKernelGGL((filt_halo_kernel<PS_DTYPE_F64>), grid, dim3(64), 0, ctx->stream, cfg, f, n, S, H, ctx->filt_fwd.as<double>(), d_out, st);
    else                                hipLaunchKernelGGL((filt_halo_kernel<PS_DTYPE_I16>), grid, dim3(64), 0, ctx->stream, cfg, f, n, S, H, ctx->filt_fwd.as<double>(), d_out, st);
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->defer_sync) return PS_OK;                   // (a batch: one status check for all of its events)
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_small.p, ctx->small.p, sizeof(SmallLayout), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return check_status(ctx, static_cast<unsigned>(ctx->h_small.as<SmallLayout>()->status));
}
}  // namespace

int ps_filter_bessel(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt, int64_t n, int32_t order,
                     double cutoff, double sampling_freq, double *d_out)
{
    if (!ctx) return PS_ERR_ARG;
    if (!d_samples || !d_out) return fail(ctx, PS_ERR_ARG, "null pointer");
    if (order < 1 || order > FILT_MAXORD)
        return fail(ctx, PS_ERR_ARG, "Bessel orders 1..%d run on the device (the reference's default is 1)", FILT_MAXORD);
    const int padlen = 3 * (order + 1);                  // scipy: 3 * max(len(a), len(b))
    if (n <= padlen) return fail(ctx, PS_ERR_ARG, "the length of the input must be greater than padlen, which is %d", padlen);
    const double wn = cutoff / (sampling_freq / 2.0);
    if (!(wn > 0.0) || !(wn < 1.0)) return fail(ctx, PS_ERR_ARG, "cutoff must lie strictly between 0 and the Nyquist frequency");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevCfg cfg;
    int rc;
    if (fmt && fmt->dtype == PS_DTYPE_F64) {
        // float64 input (this entry only): the values are the current itself
        const ps_sample_format f32 = {PS_DTYPE_F32, 0, 1.0};
        rc = make_cfg(ctx, d_samples, &f32, 1, 1, 2, 0.0, &cfg);
        if (rc) return rc;
        cfg.dtype = PS_DTYPE_F64; cfg.q = 1.0; cfg.q2 = 1.0; cfg.inv_q = 1.0f;
    } else {
        rc = make_cfg(ctx, d_samples, fmt, 1, 1, 2, 0.0, &cfg);
    }
    if (rc) return rc;
    if (order > 1) return filter_order_n(ctx, cfg, n, order, wn, d_out);
    // scipy.signal.bessel(1, wn, 'low', analog=False): one real pole, bilinear transform with pre-warping (fs = 2)
    FiltCoef f;
    const double wo = 4.0 * std::tan(3.14159265358979323846 * wn / 2.0);
    f.b0 = wo / (4.0 + wo); f.b1 = f.b0; f.a1 = (wo - 4.0) / (wo + 4.0);
    f.alpha = -f.a1; f.beta = f.b1 - f.a1 * f.b0;
    f.zi = (f.b1 - f.a1 * f.b0) / (1.0 + f.a1);        // lfilter_zi: steady state of the delay for a unit step input
    FiltGeom g;
    g.n = n; g.total = n + 2 * FILT_PAD;
    // Fast filters (the state forgets within a halo of <= 1024 samples): both directions in one kernel over tiles
    // with halos.  alpha^H <= 2^-60.
    if (ctx->filter_fused && f.alpha > 0.0 && f.alpha < 1.0) {
        const double h_req = std::ceil(60.0 * std::log(2.0) / -std::log(f.alpha));
        const int H = h_req <= 1024.0 ? std::max(64, (static_cast<int>(h_req) + 63) / 64 * 64) : 0;   // a multiple of 64, at most 1024
        if (H) {
            g.lead = 0; g.padded = g.total;
            const int T = FILT_CHUNK - 2 * H;
            const dim3 fgrid(static_cast<unsigned>((g.total + T - 1) / T));
            HIP_TRY(ctx, ctx->h_small.reserve(sizeof(SmallLayout)));
            if (!ctx->defer_sync) HIP_TRY(ctx, hipMemsetAsync(ctx->small.p, 0, sizeof(SmallLayout), ctx->stream));
            unsigned *fst = reinterpret_cast<unsigned *>(&ctx->small.as<SmallLayout>()->status);
            if (cfg.dtype == PS_DTYPE_F32)      hipLaunchKernelGGL((filt_fused_kernel<PS_DTYPE_F32>), fgrid, dim3(FILT_NT), 0, ctx->stream, cfg, f, g, H, d_out, fst);
            else if (cfg.dtype == PS_DTYPE_F64) hipLaunchKernelGGL((filt_fused_kernel<PS_DTYPE_F64>), fgrid, dim3(FILT_NT), 0, ctx->stream, cfg, f, g, H, d_out, fst);
            else                                hipLaunchKernelGGL((filt_fused_kernel<PS_DTYPE_I16>), fgrid, dim3(FILT_NT), 0, ctx->stream, cfg, f, g, H, d_out, fst);
            HIP_TRY(ctx, hipGetLastError());
            if (ctx->defer_sync) return PS_OK;
            HIP_TRY(ctx, hipMemcpyAsync(ctx->h_small.p, ctx->small.p, sizeof(SmallLayout), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            return check_status(ctx, static_cast<unsigned>(ctx->h_small.as<SmallLayout>()->status));
        }
    }
    const int64_t n_chunks = (g.total + FILT_CHUNK - 1) / FILT_CHUNK;
    g.padded = n_chunks * FILT_CHUNK;                  // the intermediate is stored behind a lead-in (seg_filter.hpp)
    g.lead = g.padded - g.total;
    HIP_TRY(ctx, ctx->filt_fwd.reserve(static_cast<size_t>(g.padded) * sizeof(double)));
    HIP_TRY(ctx, ctx->filt_agg.reserve(static_cast<size_t>(n_chunks) * 2 * sizeof(double2)));
    HIP_TRY(ctx, ctx->filt_zin.reserve(static_cast<size_t>(n_chunks) * sizeof(double)));
    HIP_TRY(ctx, ctx->h_small.reserve(sizeof(SmallLayout)));
    if (!ctx->defer_sync) HIP_TRY(ctx, hipMemsetAsync(ctx->small.p, 0, sizeof(SmallLayout), ctx->stream));
    SmallLayout *sm = ctx->small.as<SmallLayout>();
    unsigned *st = reinterpret_cast<unsigned *>(&sm->status);
    double *fwd = ctx->filt_fwd.as<double>();
    double2 *agg = ctx->filt_agg.as<double2>(), *agg_b = agg + n_chunks;
    double *zin = ctx->filt_zin.as<double>();
    const dim3 grid(static_cast<unsigned>(n_chunks));
#define PS_FILT(DT)                                                                                                          \
    hipLaunchKernelGGL((filt_local_kernel<DT>), grid, dim3(FILT_NT), 0, ctx->stream, cfg, f, g, agg, st);                     \
    hipLaunchKernelGGL((filt_carry_kernel<0, DT>), dim3(1), dim3(1024), 0, ctx->stream, cfg, f, fwd, g, agg, n_chunks, zin);   \
    hipLaunchKernelGGL((filt_apply_kernel<0, DT>), grid, dim3(FILT_NT), 0, ctx->stream, cfg, f, fwd, g, zin, fwd, agg_b, st); \
    hipLaunchKernelGGL((filt_carry_kernel<1, DT>), dim3(1), dim3(1024), 0, ctx->stream, cfg, f, fwd, g, agg_b, n_chunks, zin); \
    hipLaunchKernelGGL((filt_apply_kernel<1, DT>), grid, dim3(FILT_NT), 0, ctx->stream, cfg, f, fwd, g, zin, d_out, agg_b, st);
    if (cfg.dtype == PS_DTYPE_F32) { PS_FILT(PS_DTYPE_F32) } else if (cfg.dtype == PS_DTYPE_F64) { PS_FILT(PS_DTYPE_F64) } else { PS_FILT(PS_DTYPE_I16) }
#undef PS_FILT
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->defer_sync) return PS_OK;                   // (a batch: one status check for all of its events)
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_small.p, ctx->small.p, sizeof(SmallLayout), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return check_status(ctx, static_cast<unsigned>(ctx->h_small.as<SmallLayout>()->status));
}

int ps_requantise(ps_ctx *ctx, const double *d_in, int64_t n, float *d_out, double *centre_out, double *step_out)
{
    if (!ctx) return PS_ERR_ARG;
    if (!d_in || !d_out || !centre_out || !step_out || n < 1) return fail(ctx, PS_ERR_ARG, "null pointer or empty input");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(1024, (n + 8 * RQ_NT - 1) / (8 * RQ_NT)));
    HIP_TRY(ctx, ctx->filt_agg.reserve(static_cast<size_t>(grid) * 3 * sizeof(double)));
    HIP_TRY(ctx, ctx->h_meta.reserve(static_cast<size_t>(grid) * 3 * sizeof(double)));
    hipLaunchKernelGGL(requant_stats_kernel, dim3(grid), dim3(RQ_NT), 0, ctx->stream, d_in, static_cast<long long>(n), ctx->filt_agg.as<double>());
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_meta.p, ctx->filt_agg.p, static_cast<size_t>(grid) * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const double *part = ctx->h_meta.as<double>();
    double sum = 0.0, mn = INFINITY, mx = -INFINITY;
    for (unsigned g = 0; g < grid; ++g) { sum += part[3 * g]; mn = std::min(mn, part[3 * g + 1]); mx = std::max(mx, part[3 * g + 2]); }
    if (!std::isfinite(sum) || !std::isfinite(mn) || !std::isfinite(mx)) return fail(ctx, PS_ERR_ARG, "the current holds NaN or infinity");
    double centre = sum / static_cast<double>(n);
    const double span = std::max(mx - centre, centre - mn);              // = max |x - centre| (subtraction is monotone)
    const double step = span > 0.0 ? std::ldexp(1.0, static_cast<int>(std::ceil(std::log2(span * 1.01))) - 22) : 1.0;
    centre = std::nearbyint(centre / step) * step;
    const unsigned rg = static_cast<unsigned>(std::min<int64_t>(65535, (n + 4 * RQ_NT - 1) / (4 * RQ_NT)));
    hipLaunchKernelGGL(requant_round_kernel, dim3(rg), dim3(RQ_NT), 0, ctx->stream, d_in, static_cast<long long>(n), centre, 1.0 / step, step, d_out);
    HIP_TRY(ctx, hipGetLastError());
    // (the context's stream does not block on torch's: the caller may read d_out, or free d_in, as soon as this returns)
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *centre_out = centre; *step_out = step;
    return PS_OK;
}

int ps_filter_requantise_batch(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt, const int64_t *ev_start,
                               const int64_t *ev_len, int32_t n_ev, int32_t order, double cutoff, double sampling_freq,
                               double *d_filtered, float *d_rounded, double *h_centre, double *h_step)
{
    if (!ctx) return PS_ERR_ARG;
    if (n_ev < 0 || (n_ev > 0 && (!d_samples || !fmt || !ev_start || !ev_len || !d_filtered || !d_rounded || !h_centre || !h_step)))
        return fail(ctx, PS_ERR_ARG, "null pointer or negative count");
    if (n_ev == 0) return PS_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t es = fmt->dtype == PS_DTYPE_I16 ? 2 : fmt->dtype == PS_DTYPE_F64 ? 8 : 4;
    std::vector<int64_t> off(static_cast<size_t>(n_ev) + 1, 0);
    std::vector<int> by_len(static_cast<size_t>(n_ev));
    for (int e = 0; e < n_ev; ++e) {
        if (ev_len[e] < 1 || ev_start[e] < 0) return fail(ctx, PS_ERR_ARG, "event %d: empty or negative range", e);
        off[e + 1] = off[e] + ev_len[e];
        by_len[e] = e;
    }
    // (longest first: the scratch of the filter paths grows at most once, before anything that uses it is in flight)
    std::sort(by_len.begin(), by_len.end(), [&](int a, int b) { return ev_len[a] > ev_len[b]; });
    HIP_TRY(ctx, ctx->h_small.reserve(sizeof(SmallLayout)));
    HIP_TRY(ctx, hipMemsetAsync(ctx->small.p, 0, sizeof(SmallLayout), ctx->stream));
    // 1. every event's filter, queued back to back (one status word for all of them)
    ctx->defer_sync = true;
    int rc = PS_OK;
    for (int k = 0; k < n_ev && rc == PS_OK; ++k) {
        const int e = by_len[k];
        rc = ps_filter_bessel(ctx, static_cast<const char *>(d_samples) + static_cast<size_t>(ev_start[e]) * es, fmt, ev_len[e], order, cutoff,
                              sampling_freq, d_filtered + off[e]);
    }
    ctx->defer_sync = false;
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    // 2. sum / min / max of every filtered current, all events' partial results in one copy
    std::vector<unsigned> grid(static_cast<size_t>(n_ev));
    std::vector<size_t> poff(static_cast<size_t>(n_ev) + 1, 0);
    for (int e = 0; e < n_ev; ++e) {
        grid[e] = static_cast<unsigned>(std::min<int64_t>(1024, (ev_len[e] + 8 * RQ_NT - 1) / (8 * RQ_NT)));
        poff[e + 1] = poff[e] + grid[e];
    }
    const size_t pbytes = poff[n_ev] * 3 * sizeof(double);
    HIP_TRY(ctx, ctx->filt_agg.reserve(pbytes));         // (the filters above are done with it only in stream order: reserve() may
    HIP_TRY(ctx, ctx->h_meta.reserve(pbytes));           //  free and allocate, which waits for the device -- correct, and rare)
    for (int e = 0; e < n_ev; ++e)
        hipLaunchKernelGGL(requant_stats_kernel, dim3(grid[e]), dim3(RQ_NT), 0, ctx->stream, d_filtered + off[e], static_cast<long long>(ev_len[e]),
                           ctx->filt_agg.as<double>() + 3 * poff[e]);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_meta.p, ctx->filt_agg.p, pbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_small.p, ctx->small.p, sizeof(SmallLayout), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    rc = check_status(ctx, static_cast<unsigned>(ctx->h_small.as<SmallLayout>()->status));
    if (rc) return rc;
    // 3. centre and grid step per event (ps_requantise's rule), then the rounding passes
    const double *part = ctx->h_meta.as<double>();
    for (int e = 0; e < n_ev; ++e) {
        double sum = 0.0, mn = INFINITY, mx = -INFINITY;
        for (size_t g = poff[e]; g < poff[e + 1]; ++g) { sum += part[3 * g]; mn = std::min(mn, part[3 * g + 1]); mx = std::max(mx, part[3 * g + 2]); }
        if (!std::isfinite(sum) || !std::isfinite(mn) || !std::isfinite(mx)) return fail(ctx, PS_ERR_ARG, "the current of event %d holds NaN or infinity", e);
        double centre = sum / static_cast<double>(ev_len[e]);
        const double span = std::max(mx - centre, centre - mn);
        const double step = span > 0.0 ? std::ldexp(1.0, static_cast<int>(std::ceil(std::log2(span * 1.01))) - 22) : 1.0;
        centre = std::nearbyint(centre / step) * step;
        h_centre[e] = centre; h_step[e] = step;
        const unsigned rg = static_cast<unsigned>(std::min<int64_t>(65535, (ev_len[e] + 4 * RQ_NT - 1) / (4 * RQ_NT)));
        hipLaunchKernelGGL(requant_round_kernel, dim3(rg), dim3(RQ_NT), 0, ctx->stream, d_filtered + off[e], static_cast<long long>(ev_len[e]), centre,
                           1.0 / step, step, d_rounded + off[e]);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PS_OK;
}

int ps_align_batch(ps_ctx *ctx, const double *h_model_means, const double *h_model_stds, const double *h_model_durs,
                   int32_t m, double skip_penalty, double backslip_penalty, const double *d_seq_means,
                   const double *d_seq_stds, const double *d_seq_durs, const int64_t *h_seq_off, int32_t n_seq,
                   double *d_scores, uint32_t *d_paths, int32_t *d_status)
{
    if (!ctx) return PS_ERR_ARG;
    if (!h_model_means || !h_model_stds || !h_model_durs || !h_seq_off) return fail(ctx, PS_ERR_ARG, "null pointer");
    if (m < 1 || m > ALIGN_M_MAX) return fail(ctx, PS_ERR_ARG, "model of %d segments: the device aligner takes 1..%d", m, ALIGN_M_MAX);
    if (n_seq < 0) return fail(ctx, PS_ERR_ARG, "negative sequence count");
    if (n_seq == 0) return PS_OK;
    if (!d_scores || !d_paths || !d_status) return fail(ctx, PS_ERR_ARG, "null output pointer");
    int64_t s_max = 0;
    for (int32_t q = 0; q < n_seq; ++q) {
        const int64_t len = h_seq_off[q + 1] - h_seq_off[q];
        if (len < 0 || h_seq_off[q] < 0) return fail(ctx, PS_ERR_ARG, "sequence offsets must be non-negative and ascending");
        if (len > INT32_MAX / 2) return fail(ctx, PS_ERR_ARG, "sequence %d too long", q);
        s_max = std::max(s_max, len);
    }
    if (h_seq_off[n_seq] > 0 && (!d_seq_means || !d_seq_stds || !d_seq_durs)) return fail(ctx, PS_ERR_ARG, "null sequence pointer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // one upload: model rows (mean, std, dur*skip, dur*backslip, first-row penalty, dur), then the offsets
    const size_t model_bytes = static_cast<size_t>(6) * m * sizeof(double);
    const size_t off_bytes = (static_cast<size_t>(n_seq) + 1) * sizeof(int64_t);
    HIP_TRY(ctx, ctx->h_up.reserve(model_bytes + off_bytes));
    HIP_TRY(ctx, ctx->align_in.reserve(model_bytes + off_bytes));
    {
        double *h = ctx->h_up.as<double>();
        double run = 0.0;                                   // np.cumsum(model_dur): sequential (calignment.pyx:30)
        for (int j = 0; j < m; ++j) {
            run = j ? run + h_model_durs[j] : h_model_durs[j];
            h[j] = h_model_means[j];
            h[m + j] = h_model_stds[j];
            h[2 * m + j] = h_model_durs[j] * skip_penalty;              // :57
            h[3 * m + j] = h_model_durs[j] * backslip_penalty;          // :61-62
            h[4 * m + j] = skip_penalty * (run - h_model_durs[j]);      // :52
            h[5 * m + j] = h_model_durs[j];
        }
        std::memcpy(h + 6 * m, h_seq_off, off_bytes);
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->align_in.p, ctx->h_up.p, model_bytes + off_bytes, hipMemcpyHostToDevice, ctx->stream));
    AlignModel M;
    const double *dm = ctx->align_in.as<double>();
    M.mean = dm; M.std = dm + m; M.dsp = dm + 2 * m; M.dbp = dm + 3 * m; M.pen0 = dm + 4 * m; M.dur = dm + 5 * m;
    M.m = m; M.skip_pen = skip_penalty; M.back_pen = backslip_penalty;
    const long long *d_off = reinterpret_cast<const long long *>(dm + 6 * m);
    // scratch: score, skip_score and backslip_score of one sequence per resident workgroup
    // traceback block: as many rows as fit in ~48 KB of LDS next to the 9 working rows; 12 KB when the batch is
    // large enough to want many resident workgroups per CU instead
    const size_t blk_budget = n_seq > 512 ? (12u << 10) : (48u << 10);
    const int B = static_cast<int>(std::max<size_t>(1, std::min<size_t>(ALIGN_B_MAX, blk_budget / (3u * m * sizeof(double)))));
    const size_t lds = align_lds_doubles(m, B) * sizeof(double);
    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(align_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(lds)));
    const unsigned long long per_wg = 3ull * static_cast<unsigned long long>(std::max<int64_t>(s_max, 1)) * m;   // doubles
    const unsigned long long budget = 2ull << 30;          // bytes of scratch at most (unless one sequence needs more)
    if (per_wg * sizeof(double) > (32ull << 30)) return fail(ctx, PS_ERR_ARG, "a %lld x %d alignment needs more than 32 GiB of scratch", (long long)s_max, m);
    unsigned grid = std::min<unsigned>(static_cast<unsigned>(n_seq), resident_slots(ctx, align_kernel, ALIGN_NT, lds));
    grid = static_cast<unsigned>(std::max<unsigned long long>(1, std::min<unsigned long long>(grid, budget / (per_wg * sizeof(double)))));
    HIP_TRY(ctx, ctx->align_scratch.reserve(static_cast<size_t>(grid) * per_wg * sizeof(double)));
    hipLaunchKernelGGL(align_kernel, dim3(grid), dim3(ALIGN_NT), lds, ctx->stream, M, d_seq_means, d_seq_stds, d_seq_durs,
                       d_off, n_seq, ctx->align_scratch.as<double>(), static_cast<long long>(per_wg), B, d_scores, d_paths, d_status);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PS_OK;
}

}  // extern "C"
