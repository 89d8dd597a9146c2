// seg_align.hpp -- the segment aligner: cSegmentAligner.align of the reference
// (calignment.pyx:20-100) for a batch of sequences against one model, on gfx950.
//
// The dynamic programme has s rows (sequence segments) of m cells (model segments).  A row depends on
// the whole previous row through two running maxima -- skip_score from the left, backslip_score from the
// right (calignment.pyx:55-61) -- whose subtraction chains are NOT associative in floating point, so
// they are kept sequential to stay bit-exact with the reference (the traceback compares scores to a
// 1e-6 relative tolerance, and a different rounding could flip a decision).  The parallelism is
// therefore:  across sequences (one 64-lane wave per sequence, thousands resident), across the two
// chains (lane 0 walks left to right, lane 1 right to left, in the same loop), and across the cells of
// a row for everything else (match score, the 4-way maximum, the stores).
//
// Data: the model (means, stds, duration x penalty products, first-row penalty) and four rows live in
// LDS; the three s x m matrices the traceback needs (score, skip_score, backslip_score) stream row by
// row to a per-workgroup scratch in HBM with coalesced stores and come back in blocks of up to 32 rows during
// the traceback (block -> LDS in one round trip, then lane 0 walks it).  fp64 throughout, no FMA contraction (the reference
// build has none).
#pragma once

namespace ps {

constexpr int ALIGN_NT = 64;
constexpr int ALIGN_M_MAX = 1024;               // model segments (LDS: 13+ rows of m doubles)
constexpr int ALIGN_B_MAX = 32;                 // rows per traceback block
constexpr double ALIGN_NEGINF = -99999999.0;    // calignment.pyx:38

// per-sequence status: what the compiled reference does on the same input (include/poreseg.h)
constexpr int AL_OK = 0, AL_VALUE = 1, AL_INDEX = 2, AL_ZERODIV = 3, AL_UNDEFINED = 4;

struct AlignModel {            // device pointers, m doubles each
    const double *mean, *std, *dsp, *dbp, *pen0, *dur;   // dsp = dur*skip_penalty, dbp = dur*backslip_penalty,
    int m;                                               // pen0 = skip_penalty*(cumdur - dur)  (calignment.pyx:52)
    double skip_pen, back_pen;
};

__device__ __forceinline__ double al_max(double a, double b) { return a >= b ? a : b; }   // calignment.pyx:8

// match[i][j] * dur_i  (calignment.pyx:49 and the products at :52,:69,:75)
__device__ __forceinline__ double al_match_dur(double sm, double ss, double sd, double mm, double ms)
{
#pragma clang fp contract(off)
    const double d = sm - mm;
    const double q = -(d * d) / (ss * ms);
    return q * sd;
}

// The workgroup is one wave: LDS traffic of a wave is in order, so lanes see each other's LDS writes after a
// compiler-level fence -- no s_barrier, and no wait for the row stores that are still on their way to HBM.
__device__ __forceinline__ void al_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// LDS (doubles): 5 model rows, 2 score rows, skip, back | traceback block: (B+1) score rows, B skip rows, B back rows
// | 3 x 64 sequence values | 2 dummy cells
__host__ __device__ inline size_t align_lds_doubles(int m, int B) { return static_cast<size_t>(9 + 3 * B + 1) * m + 3 * 64 + 2; }

__global__ __launch_bounds__(ALIGN_NT) void align_kernel(AlignModel M, const double *seq_mean, const double *seq_std,
                                                         const double *seq_dur, const long long *seq_off, int n_seq,
                                                         double *scratch, long long scratch_stride,   // doubles per WG
                                                         int B, double *scores, unsigned *paths, int *status)
{
#pragma clang fp contract(off)
    extern __shared__ double al_lds[];
    const int m = M.m, lane = threadIdx.x;
    double *l_mean = al_lds, *l_std = l_mean + m, *l_dsp = l_std + m, *l_dbp = l_dsp + m, *l_pen0 = l_dbp + m;
    double *row_a = l_pen0 + m, *row_b = row_a + m, *l_skip = row_b + m, *l_back = l_skip + m;
    double *b_score = l_back + m, *b_skip = b_score + static_cast<size_t>(B + 1) * m, *b_back = b_skip + static_cast<size_t>(B) * m;
    double *q_mean = b_back + static_cast<size_t>(B) * m, *q_std = q_mean + 64, *q_dur = q_std + 64;
    __shared__ int s_flag;
    __shared__ unsigned s_j;
    for (int j = lane; j < m; j += ALIGN_NT) {
        l_mean[j] = M.mean[j]; l_std[j] = M.std[j]; l_dsp[j] = M.dsp[j]; l_dbp[j] = M.dbp[j]; l_pen0[j] = M.pen0[j];
    }
    al_sync();
    double min_model_std = __builtin_inf();
    for (int j = lane; j < m; j += ALIGN_NT) min_model_std = fmin(min_model_std, fabs(l_std[j]));
    for (int d = 32; d; d >>= 1) min_model_std = fmin(min_model_std, __shfl_xor(min_model_std, d));
    double *g_score = scratch + static_cast<long long>(blockIdx.x) * scratch_stride;

    for (int q = blockIdx.x; q < n_seq; q += gridDim.x) {
        const long long base = seq_off[q];
        const int s = static_cast<int>(seq_off[q + 1] - base);
        if (s <= 0) { if (lane == 0) { status[q] = AL_VALUE; scores[q] = 0.0; } continue; }   // :40-41 (0, m) array: ValueError
        const double *sm = seq_mean + base, *ss = seq_std + base, *sd = seq_dur + base;
        unsigned *path = paths + base;
        double *g_skip = g_score + static_cast<long long>(s) * m, *g_back = g_skip + static_cast<long long>(s) * m;
        al_sync();
        if (lane == 0) s_flag = 0;
        // seq_std * model_std == 0 anywhere: the reference raises ZeroDivisionError while it fills match (:47-49),
        // before anything else.  Rounding is monotone, so some product is zero iff the product of the smallest
        // magnitudes is.
        {
            double lo = __builtin_inf();
            for (int i = lane; i < s; i += ALIGN_NT) lo = fmin(lo, fabs(ss[i]));
            for (int d = 32; d; d >>= 1) lo = fmin(lo, __shfl_xor(lo, d));
            if (lane == 0 && lo * min_model_std == 0.0) s_flag = AL_ZERODIV;
        }
        // the first 64 segments of the sequence (the forward pass reloads every 64 rows)
        { const int r = min(lane, s - 1); q_mean[lane] = sm[r]; q_std[lane] = ss[r]; q_dur[lane] = sd[r]; }
        al_sync();
        if (s_flag) { if (lane == 0) { status[q] = s_flag; scores[q] = 0.0; } continue; }

        double *prev = row_a, *cur = row_b;
        {   // row 0 (:51-52)
            const double m0 = q_mean[0], s0 = q_std[0], d0 = q_dur[0];
            for (int j = lane; j < m; j += ALIGN_NT) {
                const double v = al_match_dur(m0, s0, d0, l_mean[j], l_std[j]) - l_pen0[j];
                cur[j] = v;
                g_score[j] = v;
            }
        }
        if (s > 1 && m < 2) {                                   // :59-62 index 1 / m-2 of a one-segment model: IndexError
            if (lane == 0) { status[q] = AL_INDEX; scores[q] = 0.0; }
            continue;
        }
        al_sync();
        for (int i = 1; i < s; ++i) {
            double *t = prev; prev = cur; cur = t;
            if ((i & 63) == 0) {                                // next 64 segments (rows i .. i+63)
                const int r = min(i + lane, s - 1);
                const double a = sm[r], b = ss[r], c = sd[r];
                al_sync();
                q_mean[lane] = a; q_std[lane] = b; q_dur[lane] = c;
                al_sync();
            }
            // the two chains (:55-61): lane 0 skip_score left to right, lane 1 backslip_score right to left
            if (lane < 2) {
                // both lanes run the same instructions: per-lane start index and direction, no branches; the batch's
                // LDS reads fly together, the chain itself is sequential; writes past the end go to a dummy cell
                constexpr int U = 8;
                const int dir = lane == 0 ? 1 : -1;
                const double *src = lane == 0 ? prev : prev + (m - 1);          // score[i-1][j-1]  |  score[i-1][j+1]
                const double *pen = lane == 0 ? l_dsp + 1 : l_dbp + (m - 1);
                double *dst = lane == 0 ? l_skip + 1 : l_back + (m - 2);
                double *dummy = q_dur + 64 + lane;
                double r = ALIGN_NEGINF;
                dst[-dir] = r;                                               // skip[0] | back[m-1]
                for (int k0 = 0; k0 < m - 1; k0 += U) {
                    double a[U], pn[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int k = min(k0 + u, m - 2) * dir;
                        a[u] = src[k];
                        pn[u] = pen[k];
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        r = al_max(r, a[u]) - pn[u];
                        a[u] = r;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int k = k0 + u;
                        double *w = k < m - 1 ? dst + k * dir : dummy;
                        *w = a[u];
                    }
                }
            }
            al_sync();
            const double mi = q_mean[i & 63], si = q_std[i & 63], di = q_dur[i & 63];
            const long long ro = static_cast<long long>(i) * m;
            for (int j = lane; j < m; j += ALIGN_NT) {                      // :63-69
                double p = prev[j];
                if (j > 0) {
                    if (prev[j - 1] > p) p = prev[j - 1];
                    if (l_skip[j - 1] > p) p = l_skip[j - 1];
                }
                if (j < m - 1) p = al_max(p, l_back[j]);
                const double v = p + al_match_dur(mi, si, di, l_mean[j], l_std[j]);
                cur[j] = v;
                g_score[ro + j] = v; g_skip[ro + j] = l_skip[j]; g_back[ro + j] = l_back[j];
            }
            al_sync();
        }

        // double_argmax of the last row (:11-18, :71): first strict maximum above -1
        {
            double best = -1.0; int bj = 0x7fffffff;
            for (int j = lane; j < m; j += ALIGN_NT) if (cur[j] > best) { best = cur[j]; bj = j; }
            for (int d = 32; d; d >>= 1) {
                const double ob = __shfl_xor(best, d); const int oj = __shfl_xor(bj, d);
                if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
            }
            if (bj == 0x7fffffff) {                                         // undefined in the reference
                if (lane == 0) { status[q] = AL_UNDEFINED; scores[q] = 0.0; }
                continue;
            }
            if (lane == 0) s_j = static_cast<unsigned>(bj);
        }
        const double last_score = cur[m - 1];                               // :100 score[s-1, m-1]

        // traceback (:73-97): blocks of B rows come back from the scratch into LDS (contiguous, one round trip per
        // block), lane 0 walks the block.  The fence waits for the row stores and invalidates the vector L1 (which may
        // hold lines of the previous sequence that used this scratch), so plain loads see this sequence's rows.
        __threadfence();
        for (int hi = s - 1; hi >= 1; hi -= B) {
            const int lo = max(1, hi - B + 1), nb = hi - lo + 1;
            al_sync();
            {
                const double *gs = g_score + static_cast<long long>(lo - 1) * m;
                const double *gk = g_skip + static_cast<long long>(lo) * m, *gb = g_back + static_cast<long long>(lo) * m;
                for (int e = lane; e < (nb + 1) * m; e += ALIGN_NT) b_score[e] = gs[e];
                for (int e = lane; e < nb * m; e += ALIGN_NT) { b_skip[e] = gk[e]; b_back[e] = gb[e]; }
                if (lane < nb) { q_mean[lane] = sm[lo + lane]; q_std[lane] = ss[lo + lane]; q_dur[lane] = sd[lo + lane]; }
            }
            al_sync();
            if (lane == 0 && !s_flag) {
                unsigned j = s_j;
                const unsigned mu = static_cast<unsigned>(m);
                for (int i = hi; i >= lo; --i) {
                    path[i] = j;
                    if (j >= mu || j == 0) { s_flag = AL_INDEX; break; }     // score[i, j] / score[i-1, j-1] out of bounds
                    const int r = i - lo;
                    const double *c_row = b_score + static_cast<size_t>(r + 1) * m, *p_row = b_score + static_cast<size_t>(r) * m;
                    const double *k_row = b_skip + static_cast<size_t>(r) * m, *bk_row = b_back + static_cast<size_t>(r) * m;
                    const double pt = c_row[j] - al_match_dur(q_mean[r], q_std[r], q_dur[r], l_mean[j], l_std[j]);
                    const double tol = 1e-6 * fabs(pt);
                    if (fabs(pt - p_row[j - 1]) <= tol) { j -= 1; continue; }
                    if (fabs(pt - p_row[j]) <= tol) continue;
                    if (j < mu - 1) {
                        unsigned k = j; double t = pt;
                        while (k < mu - 1 && fabs(t - bk_row[k]) <= tol) { k += 1; t += l_dbp[k]; }
                        if (k > j) { j = k; continue; }
                    }
                    if (j > 0) {
                        unsigned k = j; double t = pt;
                        while (k >= 1 && fabs(t - k_row[k - 1]) <= tol) { k -= 1; t += l_dsp[k]; }
                        if (k < j) { j = k - 1u; continue; }
                    }
                }
                s_j = j;
            }
            al_sync();
            if (s_flag) break;
        }
        if (lane == 0) {
            if (s_flag) { status[q] = s_flag; scores[q] = 0.0; }
            else { path[0] = s_j; status[q] = AL_OK; scores[q] = last_score; }
        }
    }
}

}  // namespace ps
