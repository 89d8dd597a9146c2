// seg_filter.hpp -- K5: Event.filter (DataTypes.py:258-274), the order-1 Bessel low-pass applied forward and
// backward (scipy.signal.filtfilt, method "pad": odd extension by padlen = 6 samples, initial state zi * first value).
//
// A first-order section in direct form II transposed is  y[i] = b0 x[i] + z,  z <- b1 x[i] - a1 y[i], i.e. the state
// obeys the linear recurrence  z <- alpha z + beta x[i]  with alpha = -a1, beta = b1 - a1 b0.  Affine maps compose, so
// the recurrence is a scan: every workgroup folds its 4 096-sample chunk into one map (filt_local_kernel), one
// workgroup runs the maps of all chunks to get the state entering each chunk (filt_carry_kernel), and a last pass
// replays the chunks from their true entry state and writes the output (filt_apply_kernel).  All arithmetic is
// fp64; the result differs from a sequential evaluation only by the re-association of the scan (~1e-16 relative).
// The pass over the extended input (forward) is followed by the same scan over the reversed intermediate (backward).
// The intermediate is stored with a lead-in that makes its length a multiple of the chunk, so that backward chunk k is
// forward chunk K-1-k in memory: the forward apply kernel, which has its chunk's output in LDS anyway, also folds it in
// reverse order into the backward map of that chunk, and the backward pass needs no separate fold (one 8 B/sample
// read less).
//
// Included by seg_device.hpp (DevCfg, load_count).
#pragma once

namespace ps {

constexpr int FILT_NT = 256;
constexpr int FILT_PER = 16;                        // consecutive samples per thread
constexpr int FILT_CHUNK = FILT_NT * FILT_PER;
constexpr int FILT_PAD = 6;                         // scipy: padlen = 3 * max(len(a), len(b)) for a first-order section

struct FiltCoef { double b0, b1, a1, alpha, beta, zi; };

// Input of the filter kernels, in the units the caller's c.q scales to pA: ADC counts for fp32 / int16 samples (grid checked,
// as everywhere), or -- PS_DTYPE_F64, this entry only -- the float64 current itself (c.q = 1): what Event.filter is handed
// when the event's current was filtered before (DataTypes.py:258-274 filters whatever self.current holds).
template <int DT>
__device__ __forceinline__ double filt_load(const DevCfg &c, int64_t idx, unsigned &bad)
{
    if constexpr (DT == PS_DTYPE_F64) return static_cast<const double *>(c.samples)[idx];
    else return static_cast<double>(load_count<DT>(c, idx, bad));
}

// Sequence element i (0 <= i < total = n + 12) of a pass.
//   PASS 0 (forward): the odd extension of the samples (in counts; scaled by the caller):  j = i - 6;
//                     j < 0: 2 x[0] - x[-j];  j >= n: 2 x[n-1] - x[2(n-1) - j];  else x[j]
//   PASS 1 (backward): the forward output read back to front; it is stored at fwd[lead + i_forward]
struct Affine { double a, s; };                     // z -> a z + s
__device__ __forceinline__ Affine after(const Affine &second, const Affine &first)      // second o first
{
    return {second.a * first.a, fma(second.a, first.s, second.s)};
}

// Exclusive scan of the per-thread maps over the workgroup (thread order); returns the map of everything before
// this thread and, in `total`, the map of the whole workgroup.
__device__ __forceinline__ Affine filt_block_exscan(Affine mine, Affine &total, Affine *wsum /*[FILT_NT/64]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Affine inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        Affine lo;
        lo.a = __shfl_up(inc.a, d); lo.s = __shfl_up(inc.s, d);
        if (lane >= d) inc = after(inc, lo);
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    Affine before = {1.0, 0.0};
    for (int w = 0; w < wave; ++w) before = after(wsum[w], before);
    total = before;
    for (int w = wave; w < FILT_NT / 64; ++w) total = after(wsum[w], total);
    // exclusive for this thread: everything before the wave, then the lanes below
    Affine ex;
    ex.a = __shfl_up(inc.a, 1); ex.s = __shfl_up(inc.s, 1);
    if (lane == 0) ex = {1.0, 0.0};
    __syncthreads();
    return after(ex, before);
}

// LDS image of a chunk: element e at e + e/FILT_PER (one pad per thread run: few conflicts when threads read
// consecutive values, coalesced global accesses on the other side).
constexpr int FILT_LDS = FILT_CHUNK + FILT_CHUNK / FILT_PER;
__device__ __forceinline__ int filt_slot(int e) { return e + e / FILT_PER; }

// Geometry of a pass.  Chunk space u = chunk * FILT_CHUNK + e; sequence index i = u - lead (forward: lead >= 0 pads the
// front so that total + lead is a multiple of the chunk; backward: lead = 0, the padding falls behind the sequence).
struct FiltGeom { int64_t n, total, lead, padded; };      // padded = total + (forward lead) = n_chunks * FILT_CHUNK

// Coalesced load of the workgroup's chunk into LDS (all loads issued before the first use: indices are clamped /
// reflected with selects, the odd extension is applied to the loaded values), then the thread's consecutive samples
// x[] and the map they apply to the state.  Elements outside the sequence are identity.
template <int PASS, int DT>
__device__ __forceinline__ Affine filt_thread_map(const DevCfg &c, const FiltCoef &f, const double *fwd, const FiltGeom &g,
                                                  int64_t chunk0, double *lds, double *x, unsigned &bad)
{
    const int64_t lead = PASS == 0 ? g.lead : 0;
    double v[FILT_PER];
    double x_first = 0.0, x_last = 0.0;
    if (PASS == 0) {
        x_first = filt_load<DT>(c, 0, bad);
        x_last = filt_load<DT>(c, g.n - 1, bad);
    }
#pragma unroll
    for (int k = 0; k < FILT_PER; ++k) {
        // the backward pass walks memory downwards: lane order is flipped there so that a wave still reads ascending addresses
        const int e = PASS == 0 ? k * FILT_NT + threadIdx.x : FILT_CHUNK - 1 - (k * FILT_NT + threadIdx.x);
        const int64_t i = min(max(chunk0 + e - lead, static_cast<int64_t>(0)), g.total - 1);
        if (PASS == 1) v[k] = fwd[g.padded - 1 - i];
        else {
            const int64_t j = i - FILT_PAD;
            const int64_t idx = j < 0 ? -j : (j >= g.n ? 2 * (g.n - 1) - j : j);
            v[k] = filt_load<DT>(c, idx, bad);
        }
    }
#pragma unroll
    for (int k = 0; k < FILT_PER; ++k) {
        const int e = PASS == 0 ? k * FILT_NT + threadIdx.x : FILT_CHUNK - 1 - (k * FILT_NT + threadIdx.x);
        double val = v[k];
        if (PASS == 0) {
            const int64_t j = chunk0 + e - lead - FILT_PAD;
            val = (j < 0 ? 2.0 * x_first - val : (j >= g.n ? 2.0 * x_last - val : val)) * c.q;
        }
        lds[filt_slot(e)] = val;
    }
    __syncthreads();
    Affine m = {1.0, 0.0};
    const int e0 = threadIdx.x * FILT_PER;
#pragma unroll
    for (int k = 0; k < FILT_PER; ++k) {
        x[k] = lds[filt_slot(e0 + k)];
        const int64_t i = chunk0 + e0 + k - lead;
        if (i >= 0 && i < g.total) { m.s = fma(f.alpha, m.s, f.beta * x[k]); m.a *= f.alpha; }
    }
    return m;
}

// Forward fold: agg[chunk] = map of the chunk.
template <int DT>
__global__ __launch_bounds__(FILT_NT) void filt_local_kernel(DevCfg c, FiltCoef f, FiltGeom g, double2 *agg, unsigned *status)
{
    __shared__ Affine wsum[FILT_NT / 64];
    __shared__ double lds[FILT_LDS];
    const int64_t chunk = blockIdx.x;
    unsigned bad = 0;
    double x[FILT_PER];
    const Affine mine = filt_thread_map<0, DT>(c, f, nullptr, g, chunk * FILT_CHUNK, lds, x, bad);
    Affine all;
    (void)filt_block_exscan(mine, all, wsum);
    if (threadIdx.x == 0) agg[chunk] = make_double2(all.a, all.s);
    if (bad) atomicOr(status, bad);
}

// zin[chunk] = state entering the chunk, chunks taken in pass order.  The entry state of the pass is zi * (its first
// element): x_ext[0] = 2 x[0] - x[6] for the forward pass, the last forward output for the backward pass.
template <int PASS, int DT>
__global__ __launch_bounds__(1024) void filt_carry_kernel(DevCfg c, FiltCoef f, const double *fwd, FiltGeom g,
                                                          const double2 *__restrict__ agg, int64_t n_chunks, double *__restrict__ zin)
{
    unsigned bad = 0;
    const double first = PASS == 1 ? fwd[g.padded - 1]
                                   : (2.0 * filt_load<DT>(c, 0, bad) - filt_load<DT>(c, FILT_PAD, bad)) * c.q;
    const double z0 = f.zi * first;
    const int64_t per = (n_chunks + 1023) / 1024;
    const int64_t c0 = static_cast<int64_t>(threadIdx.x) * per, c1 = min(n_chunks, c0 + per);
    constexpr int B = 8;                               // maps loaded per trip (independent loads, one round trip)
    Affine m = {1.0, 0.0};
    for (int64_t k0 = c0; k0 < c1; k0 += B) {
        double2 q[B];
#pragma unroll
        for (int i = 0; i < B; ++i) q[i] = agg[min(k0 + i, n_chunks - 1)];
#pragma unroll
        for (int i = 0; i < B; ++i) if (k0 + i < c1) m = after({q[i].x, q[i].y}, m);
    }
    // exclusive scan of the 1 024 thread maps: wave scans, then the 16 wave totals serially
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        Affine inc = m;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            Affine lo;
            lo.a = __shfl_up(inc.a, d); lo.s = __shfl_up(inc.s, d);
            if (lane >= d) inc = after(inc, lo);
        }
        __shared__ Affine wtot[16];
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        Affine before = {1.0, 0.0};
        for (int w = 0; w < wave; ++w) before = after(wtot[w], before);
        Affine ex;
        ex.a = __shfl_up(inc.a, 1); ex.s = __shfl_up(inc.s, 1);
        if (lane == 0) ex = {1.0, 0.0};
        m = after(ex, before);                         // everything before this thread's chunks
    }
    double z = fma(m.a, z0, m.s);
    for (int64_t k0 = c0; k0 < c1; k0 += B) {
        double2 q[B];
#pragma unroll
        for (int i = 0; i < B; ++i) q[i] = agg[min(k0 + i, n_chunks - 1)];
#pragma unroll
        for (int i = 0; i < B; ++i) if (k0 + i < c1) { zin[k0 + i] = z; z = fma(q[i].x, z, q[i].y); }
    }
}

// Replays a chunk from its true entry state.
//   PASS 0 writes the forward output to fwd_out[u] (chunk space, i.e. behind the lead-in) and folds the chunk's output,
//          taken back to front, into agg_b[K-1-chunk]: the map of the backward pass's chunk that covers the same memory.
//   PASS 1 writes out[j], j = 0..n-1 (the extension is dropped and the order restored); its chunks are taken in
//          descending order so that consecutive workgroups walk memory upwards.
template <int PASS, int DT>
__global__ __launch_bounds__(FILT_NT) void filt_apply_kernel(DevCfg c, FiltCoef f, const double *fwd, FiltGeom g, const double *zin,
                                                             double *out, double2 *agg_b, unsigned *status)
{
    __shared__ Affine wsum[FILT_NT / 64];
    __shared__ double lds[FILT_LDS];
    const int64_t chunk = PASS == 0 ? blockIdx.x : gridDim.x - 1 - blockIdx.x;
    const int64_t chunk0 = chunk * FILT_CHUNK;
    const int64_t lead = PASS == 0 ? g.lead : 0;
    unsigned bad = 0;
    double x[FILT_PER];
    const Affine mine = filt_thread_map<PASS, DT>(c, f, fwd, g, chunk0, lds, x, bad);
    Affine all;
    const Affine before = filt_block_exscan(mine, all, wsum);
    double z = fma(before.a, zin[chunk], before.s);
    const int e0 = threadIdx.x * FILT_PER;
#pragma unroll
    for (int k = 0; k < FILT_PER; ++k) {
        const int64_t i = chunk0 + e0 + k - lead;
        if (i >= 0 && i < g.total) {                   // (identity elements leave the state alone)
            const double y = fma(f.b0, x[k], z);
            z = fma(-f.a1, y, f.b1 * x[k]);
            lds[filt_slot(e0 + k)] = y;                // (each thread overwrites its own slots)
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < FILT_PER; ++k) {               // coalesced write-out (ascending addresses in both passes)
        const int e = PASS == 0 ? k * FILT_NT + threadIdx.x : FILT_CHUNK - 1 - (k * FILT_NT + threadIdx.x);
        const int64_t i = chunk0 + e - lead;
        if (i >= 0 && i < g.total) {
            const double y = lds[filt_slot(e)];
            if (PASS == 0) out[chunk0 + e] = y;
            else {
                const int64_t j = g.total - 1 - i - FILT_PAD;    // position in the original order
                if (j >= 0 && j < g.n) out[j] = y;
            }
        }
    }
    if (PASS == 0) {
        // backward fold of this chunk: thread t takes the run of thread FILT_NT-1-t back to front
        const int r0 = (FILT_NT - 1 - static_cast<int>(threadIdx.x)) * FILT_PER;
        Affine mb = {1.0, 0.0};
#pragma unroll
        for (int k = FILT_PER - 1; k >= 0; --k) {
            const int64_t i = chunk0 + r0 + k - lead;
            if (i >= 0 && i < g.total) { mb.s = fma(f.alpha, mb.s, f.beta * lds[filt_slot(r0 + k)]); mb.a *= f.alpha; }
        }
        Affine allb;
        (void)filt_block_exscan(mb, allb, wsum);
        if (threadIdx.x == 0) agg_b[gridDim.x - 1 - chunk] = make_double2(allb.a, allb.s);
        if (bad) atomicOr(status, bad);
    }
}

// ---- fused variant: both directions in one kernel, tiles with halos ---------------------------------------------
// A one-pole filter forgets: an error of the state decays by alpha per sample, so a tile that starts H samples early
// with a guessed state (the steady state of its first value) has the exact state -- to below one ulp of the output
// when alpha^H <= 2^-60 -- by the time it reaches the samples it is responsible for.  A workgroup loads T + 2H = 4 096
// elements of the extended sequence, runs the forward recurrence over all of them in LDS (exact entry state when the
// window contains the start of the sequence), the backward recurrence from the right edge (exact when the window
// contains the end), and writes the T elements in the middle: 4 096 / T x 4 B in, 8 B out per sample instead of the
// 32 B of the three-pass scan, no intermediate in HBM, no carry kernels.  The host picks H from the pole (H <= 1024,
// i.e. poles up to 0.96: cutoffs from ~0.6 kHz at 100 kHz sampling); slower filters take the exact scan above.
#ifndef PS_FILT_OCC
#define PS_FILT_OCC 2          // workgroups per CU the fused kernel is compiled for
#endif
template <int DT>
__global__ __launch_bounds__(FILT_NT, PS_FILT_OCC) void filt_fused_kernel(DevCfg c, FiltCoef f, FiltGeom g, int H, double *out, unsigned *status)
{
    __shared__ Affine wsum[FILT_NT / 64];
    __shared__ double lds[FILT_LDS];
    const int T = FILT_CHUNK - 2 * H;
    const int64_t w0 = static_cast<int64_t>(blockIdx.x) * T - H;      // window element e <-> sequence index i = w0 + e
    unsigned bad = 0;
    double x[FILT_PER];
    FiltGeom g0 = g;
    g0.lead = 0;
    // forward over the whole window
    const Affine mine = filt_thread_map<0, DT>(c, f, nullptr, g0, w0, lds, x, bad);
    const int e_first = w0 < 0 ? static_cast<int>(-w0) : 0;           // first element inside the sequence
    const double z_in = f.zi * lds[filt_slot(e_first)];               // exact at the start of the sequence (zi * x_ext[0]), a guess elsewhere
    Affine all;
    const Affine before = filt_block_exscan(mine, all, wsum);
    double z = fma(before.a, z_in, before.s);
    const int e0 = threadIdx.x * FILT_PER;
#pragma unroll
    for (int k = 0; k < FILT_PER; ++k) {
        const int64_t i = w0 + e0 + k;
        if (i >= 0 && i < g.total) {
            const double y = fma(f.b0, x[k], z);
            z = fma(-f.a1, y, f.b1 * x[k]);
            lds[filt_slot(e0 + k)] = y;
        }
    }
    __syncthreads();
    // backward over the forward output, from the right edge of the window
    const int64_t i_last = min(g.total - 1, w0 + FILT_CHUNK - 1);
    const double zb_in = f.zi * lds[filt_slot(static_cast<int>(i_last - w0))];   // exact at the end of the sequence
    const int r0 = (FILT_NT - 1 - static_cast<int>(threadIdx.x)) * FILT_PER;     // thread t: the run of thread NT-1-t, back to front
    Affine mb = {1.0, 0.0};
#pragma unroll
    for (int k = FILT_PER - 1; k >= 0; --k) {
        x[k] = lds[filt_slot(r0 + k)];
        const int64_t i = w0 + r0 + k;
        if (i >= 0 && i < g.total) { mb.s = fma(f.alpha, mb.s, f.beta * x[k]); mb.a *= f.alpha; }
    }
    Affine allb;
    const Affine before_b = filt_block_exscan(mb, allb, wsum);        // (its barriers also order the reads above before the writes below)
    z = fma(before_b.a, zb_in, before_b.s);
#pragma unroll
    for (int k = FILT_PER - 1; k >= 0; --k) {
        const int64_t i = w0 + r0 + k;
        if (i >= 0 && i < g.total) {
            const double y = fma(f.b0, x[k], z);
            z = fma(-f.a1, y, f.b1 * x[k]);
            lds[filt_slot(r0 + k)] = y;
        }
    }
    __syncthreads();
    // the tile's own elements, coalesced; sample j = i - PAD
#pragma unroll
    for (int k = 0; k < FILT_PER; ++k) {
        const int e = k * FILT_NT + threadIdx.x;
        const int64_t j = w0 + e - FILT_PAD;
        if (e >= H && e < H + T && j >= 0 && j < g.n) out[j] = lds[filt_slot(e)];
    }
    if (bad) atomicOr(status, bad);
}

// ---- orders 2..4: Event.filter(order=n) (DataTypes.py:258-274 hands `order` to scipy.signal.bessel) -------------------
// The n-th order section in direct form II transposed carries n delays:  y = b0 x + z0,  z_k <- b_{k+1} x - a_{k+1} y
// + z_{k+1}.  Orders other than 1 are rare in the reference's workflows (its default and every example use 1), so this
// path is simple rather than fast: the state of a stable filter forgets (||A^H|| <= 2^-70 for the halo H the host
// picks from the filter's own matrix), hence every THREAD filters one segment of S outputs of the extended sequence on
// its own -- forward from H samples before the segment (from the steady state of its first value; the exact zi x_ext[0]
// at the start of the sequence) to H samples behind it, the forward values kept in a private scratch row, then
// backward from the end of that row (exact zi * last value at the end of the sequence).  The arithmetic per sample is
// scipy's lfilter's, in its order; only the warm-up replaces the history.  Work: (S + 2H) + (S + H) samples per S.
constexpr int FILT_MAXORD = 8;
struct FiltN { int order, pad; double b[FILT_MAXORD + 1], a[FILT_MAXORD + 1], zi[FILT_MAXORD]; };

template <int DT>
__global__ __launch_bounds__(64) void filt_halo_kernel(DevCfg c, FiltN f, int64_t n, int S, int H, double *scratch, double *out,
                                                       unsigned *status)
{
    const int64_t m = n + 2LL * f.pad;
    const int64_t t = blockIdx.x * 64LL + threadIdx.x;
    const int64_t lo = t * S;
    if (lo >= m) return;
    const int64_t hi = min(m, lo + S);
    const int64_t fs = max(static_cast<int64_t>(0), lo - H), fe = min(m, hi + H);
    unsigned bad = 0;
    const double x0 = filt_load<DT>(c, 0, bad), xl = filt_load<DT>(c, n - 1, bad);
    auto x_ext = [&](int64_t j) {
        const int64_t jj = j - f.pad;
        const int64_t idx = jj < 0 ? -jj : (jj >= n ? 2 * (n - 1) - jj : jj);
        const double v = filt_load<DT>(c, idx, bad);
        return (jj < 0 ? 2.0 * x0 - v : (jj >= n ? 2.0 * xl - v : v)) * c.q;
    };
    double *row = scratch + t * static_cast<int64_t>(S + H);
    double z[FILT_MAXORD];
    auto step = [&](double x) {                        // scipy's lfilter, in its operation order, no FMA contraction
#pragma clang fp contract(off)
        const double y = z[0] + f.b[0] * x;
#pragma unroll
        for (int k = 0; k < FILT_MAXORD - 1; ++k) {
            const double mid = (z[k + 1] + x * f.b[k + 1]) - y * f.a[k + 1];
            const double last = x * f.b[k + 1] - y * f.a[k + 1];
            z[k] = k + 1 < f.order ? mid : (k + 1 == f.order ? last : 0.0);
        }
        z[FILT_MAXORD - 1] = f.order == FILT_MAXORD ? x * f.b[FILT_MAXORD] - y * f.a[FILT_MAXORD] : 0.0;
        return y;
    };
    {
        const double xs = x_ext(fs);
#pragma unroll
        for (int k = 0; k < FILT_MAXORD; ++k) z[k] = k < f.order ? f.zi[k] * xs : 0.0;
    }
    for (int64_t j = fs; j < fe; ++j) {
        const double y = step(x_ext(j));
        if (j >= lo) row[j - lo] = y;
    }
    {
        const double ys = row[fe - 1 - lo];
#pragma unroll
        for (int k = 0; k < FILT_MAXORD; ++k) z[k] = k < f.order ? f.zi[k] * ys : 0.0;
    }
    for (int64_t j = fe - 1; j >= lo; --j) {
        const double y = step(row[j - lo]);
        const int64_t o = j - f.pad;
        if (j < hi && o >= 0 && o < n) out[o] = y;
    }
    if (bad) atomicOr(status, bad);
}

// ---- K5b: a filtered current back onto a grid (ps_requantise) -----------------------------------------------------
// The segmenter works on exact integer sums; a filtered current is float64 off every grid.  DataTypes.Event.parse
// centres it on its mean and rounds it to the finest power-of-two grid that keeps every count below 2^22 (DESIGN.md
// 7c).  These two kernels do that for a current that is still on the device: partial sum / min / max per workgroup
// (combined on the host in a fixed order), then out[i] = (x[i] - centre) rounded to a multiple of `step`, as fp32
// (22-bit counts times a power of two: exact).  8 B in, 4 B out per sample and pass: bound by HBM.
constexpr int RQ_NT = 256;
__global__ __launch_bounds__(RQ_NT) void requant_stats_kernel(const double *x, long long n, double *part)
{
    __shared__ double s_sum[RQ_NT / 64], s_min[RQ_NT / 64], s_max[RQ_NT / 64];
    double sum = 0.0, mn = INFINITY, mx = -INFINITY;
    for (long long i = blockIdx.x * static_cast<long long>(RQ_NT) + threadIdx.x; i < n; i += gridDim.x * static_cast<long long>(RQ_NT)) {
        const double v = x[i];
        sum += v; mn = fmin(mn, v); mx = fmax(mx, v);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        sum += __shfl_down(sum, d); mn = fmin(mn, __shfl_down(mn, d)); mx = fmax(mx, __shfl_down(mx, d));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_sum[wave] = sum; s_min[wave] = mn; s_max[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < RQ_NT / 64; ++w) { sum += s_sum[w]; mn = fmin(mn, s_min[w]); mx = fmax(mx, s_max[w]); }
        part[3 * blockIdx.x] = sum; part[3 * blockIdx.x + 1] = mn; part[3 * blockIdx.x + 2] = mx;
    }
}
__global__ __launch_bounds__(RQ_NT) void requant_round_kernel(const double *x, long long n, double centre, double inv_step, double step,
                                                           float *out)
{
#pragma clang fp contract(off)
    for (long long i = blockIdx.x * static_cast<long long>(RQ_NT) + threadIdx.x; i < n; i += gridDim.x * static_cast<long long>(RQ_NT))
        out[i] = static_cast<float>(rint((x[i] - centre) * inv_step) * step);
}

}  // namespace ps
