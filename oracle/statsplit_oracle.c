/*
 * statsplit_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference change-point segmenter
 * (PyPore/cparsers.pyx, class FastStatSplit).  It exists so that tests/, the smoke
 * check and bench.py's cpu_baseline leg have something to compare the HIP path
 * against on machines where /root/reference is absent (the GPU box).  Nothing under
 * pypore_amd/ may import, link or call it.
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks every function here against
 * golden vectors recorded from the compiled, unmodified reference
 * (tests/golden/make_golden.py, oracle/build_reference.sh), G1..G8 of SURVEY.md 8(c).
 *
 * Arithmetic follows the C that Cython generates from the reference line by line:
 * fp64 throughout, sequential prefix sums (numpy add.accumulate), pow(x,2.0) for **2,
 * libm log, no FMA contraction (build with -ffp-contract=off, no -march).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- cparsers.pyx:55-101  FastStatSplit.__init__ (min_gain only) ------------------
 * "Not given" follows Python truthiness: None and 0 are both falsy in the reference,
 * so callers pass 0.0 for an omitted parameter.  Returns 0, or -1/-2/-3 for the three
 * reference assertions (cparsers.pyx:69-76). */
int so_min_gain(int min_width, int max_width, int window_width,
                double min_gain_per_sample, double false_positive_rate,
                double prior_segments_per_second, double sampling_freq,
                double cutoff_freq, double *out)
{
    double fpr = false_positive_rate, sps = prior_segments_per_second, mg;
    if (!(fpr != 0.0)) fpr = sampling_freq;               /* :64-65 */
    if (!(sps != 0.0)) sps = sampling_freq / 2.;          /* :66-67 */
    if (!(max_width >= min_width)) return -1;             /* :69 */
    if (!(window_width >= 2 * min_width)) return -2;      /* :71 */
    if (cutoff_freq != 0.0 && !(cutoff_freq <= 0.5 * sampling_freq)) return -3; /* :74-76 */
    if (min_gain_per_sample != 0.0) {
        mg = min_gain_per_sample * window_width;          /* :82-84 */
    } else {
        double k = (cutoff_freq != 0.0) ? cutoff_freq / (0.5 * sampling_freq) : 1.0; /* :89 */
        mg = (-log(sps / (sampling_freq - sps)) - log(fpr / sampling_freq)) / k;      /* :95-97 */
    }
    *out = mg * 2;                                        /* :101 */
    return 0;
}

/* ---- cparsers.pyx:110-111  c = cumsum(x), c2 = cumsum(x*x)  (sequential fp64) ---- */
static void prefix_sums(const double *x, long n, double *c, double *c2)
{
    double a = 0.0, b = 0.0;
    for (long i = 0; i < n; ++i) {
        double v = x[i];
        double sq = v * v;          /* np.multiply(current, current) */
        if (i == 0) { a = v; b = sq; } else { a = a + v; b = b + sq; }
        c[i] = a; c2[i] = b;
    }
}

/* ---- cparsers.pyx:31-38  var_c ---------------------------------------------------- */
static inline double var_c(int start, int end, const double *c, const double *c2)
{
    if (start == end) return 0;
    if (start == 0)
        return c2[end - 1] / end - pow(c[end - 1] / end, 2.0);
    return (c2[end - 1] - c2[start - 1]) / (end - start)
         - pow((c[end - 1] - c[start - 1]) / (end - start), 2.0);
}

typedef struct {
    const double *c, *c2;
    int min_width, max_width, window_width;
    double min_gain;
    long long n_evals;      /* candidate evaluations (work-amplification statistic) */
    long long n_windows;
} so_ctx;

/* ---- cparsers.pyx:157-178  _best_split_stepwise ----------------------------------- */
static int best_split_stepwise(so_ctx *k, int start, int end, double *scores)
{
    if (end - start <= 2 * k->min_width) return -1;                       /* :164 */
    double var_summed = (end - start) * log(var_c(start, end, k->c, k->c2)); /* :166 */
    double min_gain = k->min_gain;
    int x = -1;
    k->n_windows++;
    for (int i = start + k->min_width; i < end + 1 - k->min_width; ++i) {  /* :171 */
        double low = (i - start) * log(var_c(start, i, k->c, k->c2));
        double high = (end - i) * log(var_c(i, end, k->c, k->c2));
        double gain = var_summed - (low + high);
        if (scores) scores[i] = gain;                                     /* :248 */
        if (gain > min_gain) { min_gain = gain; x = i; }                  /* :175-177 */
        k->n_evals++;
    }
    return x;
}

/* Work items for the explicit-stack in-order traversal of _recursive_split. */
typedef struct { int kind, a, b; } so_item;    /* kind 0: CALL(a,b)  1: EMIT(a) */
typedef struct { so_item *v; long n, cap; } so_stack;
static int push(so_stack *s, int kind, int a, int b)
{
    if (s->n == s->cap) {
        long nc = s->cap ? s->cap * 2 : 256;
        so_item *nv = (so_item *)realloc(s->v, nc * sizeof(so_item));
        if (!nv) return -1;
        s->v = nv; s->cap = nc;
    }
    s->v[s->n].kind = kind; s->v[s->n].a = a; s->v[s->n].b = b; s->n++;
    return 0;
}

/* ---- cparsers.pyx:180-203  _recursive_split, same output order, no C recursion ---- */
static long recursive_split(so_ctx *k, int start0, int end0, int *out, long cap, unsigned char *flags)
{
    so_stack st = {0, 0, 0};
    long cnt = 0;
    const int mw = k->min_width, maxw = k->max_width, W = k->window_width;
    push(&st, 0, start0, end0);
    while (st.n) {
        so_item it = st.v[--st.n];
        if (it.kind == 1) { if (cnt < cap) { out[cnt] = it.a; if (flags) flags[cnt] = (unsigned char)it.b; } cnt++; continue; }
        int start = it.a, end = it.b, split_at = -1, forced_early = 0;
        for (long ps = start; ps < (long)end - 2 * mw; ps += W / 2) {      /* :188 */
            if (ps > (long)start + maxw) {                                 /* :189-191 */
                int a = start + maxw, b = end - mw;
                split_at = a <= b ? a : b;
                forced_early = 1;
                break;
            }
            long pe = ps + W; if (pe > end) pe = end;                      /* :193 */
            split_at = best_split_stepwise(k, (int)ps, (int)pe, 0);        /* :194 */
            if (split_at >= 0) break;                                      /* :195-196 */
        }
        if (forced_early) {                   /* [split] + rec(split, end): right side only */
            push(&st, 0, split_at, end);
            push(&st, 1, split_at, end == end0);
            continue;
        }
        if (split_at == -1) {                                              /* :198-201 */
            if (end - start <= maxw) continue;
            int a = start + maxw, b = end - mw;
            split_at = a <= b ? a : b;
        }
        push(&st, 0, split_at, end);          /* rec(start,split) + [split] + rec(split,end) */
        push(&st, 1, split_at, end == end0);  /* spine anchor: found by a frame rec(a, END) of the top-level chain */
        push(&st, 0, start, split_at);
    }
    free(st.v);
    return cnt;
}

/* ---- cparsers.pyx:103-118  FastStatSplit.parse -> breakpoints -----------------------
 * Writes the sorted breakpoint list (excluding 0 and n) to out[0..cap); returns its
 * length (may exceed cap: call again with a larger buffer), or -1 on allocation failure.
 * stats (nullable): [0] candidate evaluations, [1] window scans. */
long so_parse(const double *x, int n, int min_width, int max_width, int window_width,
              double min_gain, int *out, long cap, long long *stats)
{
    double *c = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double *c2 = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (!c || !c2) { free(c); free(c2); return -1; }
    prefix_sums(x, n, c, c2);
    so_ctx k = { c, c2, min_width, max_width, window_width, min_gain, 0, 0 };
    long cnt = recursive_split(&k, 0, n, out, cap, 0);
    if (stats) { stats[0] = k.n_evals; stats[1] = k.n_windows; }
    free(c); free(c2);
    return cnt;
}

/* so_parse plus flags[i] = 1 when breakpoint i was found by the top-level chain of right
 * recursions rec(a, n) (a "spine anchor"); test counterpart of ps_segment_batch_ex. */
long so_parse_flags(const double *x, int n, int min_width, int max_width, int window_width,
                    double min_gain, int *out, unsigned char *flags, long cap)
{
    double *c = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double *c2 = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (!c || !c2) { free(c); free(c2); return -1; }
    prefix_sums(x, n, c, c2);
    so_ctx k = { c, c2, min_width, max_width, window_width, min_gain, 0, 0 };
    long cnt = recursive_split(&k, 0, n, out, cap, flags);
    free(c); free(c2);
    return cnt;
}

/* ---- cparsers.pyx:120-155  best_single_split ------------------------------------------
 * start=0, end=len(c)-1 (sic), i in range(2, end-2), threshold 0. */
int so_best_single_split(const double *x, int n, double *gain_out, int *idx_out)
{
    double *c = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double *c2 = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (!c || !c2) { free(c); free(c2); return -1; }
    prefix_sums(x, n, c, c2);
    int end = n - 1, xbest = -1;
    double min_gain = 0.;
    double var_summed = end * log(var_c(0, end, c, c2));
    for (int i = 2; i < end - 2; ++i) {
        double low = i * log(var_c(0, i, c, c2));
        double high = (end - i) * log(var_c(i, end, c, c2));
        double gain = var_summed - (low + high);
        if (gain > min_gain) { min_gain = gain; xbest = i; }
    }
    *gain_out = min_gain; *idx_out = xbest;
    free(c); free(c2);
    return 0;
}

/* ---- cparsers.pyx:205-249  score_samples(current, no_split=True) ----------------------
 * One scan of the whole array as a single window: scores[i] = gain(i) for candidates,
 * 0 elsewhere (np.zeros); returns the split index or -1. */
int so_score_window(const double *x, int n, int min_width, double min_gain, double *scores)
{
    double *c = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double *c2 = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (!c || !c2) { free(c); free(c2); return -2; }
    prefix_sums(x, n, c, c2);
    memset(scores, 0, sizeof(double) * (size_t)n);
    so_ctx k = { c, c2, min_width, 0, 0, min_gain, 0, 0 };
    int r = best_split_stepwise(&k, 0, n, scores);
    free(c); free(c2);
    return r;
}

/* ---- core.py:209-223  Segment.mean / std(population) / min / max ------------------------
 * Straight fp64 two-pass statistics; the test tolerance for mean/std is 1e-5 relative
 * (numpy's pairwise summation differs in the last bits only). */
void so_segment_stats(const double *x, const int *bounds, long nb, int n, double *stats4)
{
    for (long s = 0; s <= nb; ++s) {
        int a = s == 0 ? 0 : bounds[s - 1], b = s == nb ? n : bounds[s];
        double sum = 0, mn = INFINITY, mx = -INFINITY;
        for (int i = a; i < b; ++i) { sum += x[i]; if (x[i] < mn) mn = x[i]; if (x[i] > mx) mx = x[i]; }
        double mean = b > a ? sum / (b - a) : NAN, ss = 0;
        for (int i = a; i < b; ++i) ss += (x[i] - mean) * (x[i] - mean);
        stats4[4 * s + 0] = mean;
        stats4[4 * s + 1] = b > a ? sqrt(ss / (b - a)) : NAN;
        stats4[4 * s + 2] = mn; stats4[4 * s + 3] = mx;
    }
}

/* ---- parsers.py:142-155 + :133-140  lambda_event_parser.parse with the default rules ----
 * mask = x < threshold; tics at mask edges; one piece per [tics[i], tics[i+1]); keep a piece
 * iff duration > min_duration and min > min_current and max < threshold.
 * Writes (start, length) pairs; returns the number of kept events. */
long so_lambda_events(const double *x, long n, double threshold, long min_duration,
                      double min_current, long *starts, long *lengths, long cap)
{
    long cnt = 0, a = 0;
    for (long i = 1; i <= n; ++i) {
        int edge = (i == n) || ((x[i] < threshold) != (x[i - 1] < threshold));
        if (!edge) continue;
        if (n > 0) {
            double mn = INFINITY, mx = -INFINITY;
            for (long j = a; j < i; ++j) { if (x[j] < mn) mn = x[j]; if (x[j] > mx) mx = x[j]; }
            if ((i - a) > min_duration && mn > min_current && mx < threshold) {
                if (cnt < cap) { starts[cnt] = a; lengths[cnt] = i - a; }
                cnt++;
            }
        }
        a = i;
    }
    return cnt;
}

/* ---- Event.filter, DataTypes.py:258-274 (order 1) ------------------------------------------------------------
 * (b, a) = scipy.signal.bessel(1, cutoff / (second / 2), btype='low', analog=0, output='ba'); current =
 * scipy.signal.filtfilt(b, a, current).  scipy is a third-party dependency of the reference (setup.py asks for
 * scipy, no pin; 1.15.3 is installed where the golden vectors were recorded); its published algorithm, restated:
 *   bessel, N = 1: analog prototype pole -1, gain 1; low-pass to the pre-warped frequency wo = 2 fs tan(pi Wn / fs)
 *   with fs = 2; bilinear transform  =>  b = [wo, wo] / (4 + wo), a = [1, (wo - 4) / (wo + 4)].
 *   filtfilt (method "pad", padtype "odd", padlen = 3 max(len a, len b) = 6): ext = odd extension of x by 6
 *   samples at both ends; zi = lfilter_zi(b, a) = (b1 - a1 b0) / (1 + a1); forward lfilter over ext with initial
 *   delay zi * ext[0]; lfilter over the reversed result with initial delay zi * (its first value); reverse; drop
 *   the extension.  lfilter (direct form II transposed): y = b0 x + z; z = b1 x - a1 y.
 * Returns 0, or -1 when n <= 6 (scipy raises ValueError) or the cutoff is not inside (0, Nyquist). */
int so_bessel1_filtfilt(const double *x, long n, double cutoff, double second, double *out)
{
    const double wn = cutoff / (second / 2.0);
    if (n <= 6 || !(wn > 0.0) || !(wn < 1.0)) return -1;
    const double wo = 4.0 * tan(3.14159265358979323846 * wn / 2.0);
    const double b0 = wo / (4.0 + wo), b1 = b0, a1 = (wo - 4.0) / (wo + 4.0);
    const double zi = (b1 - a1 * b0) / (1.0 + a1);
    const long m = n + 12;
    double *ext = (double *)malloc(sizeof(double) * (size_t)m);
    if (!ext) return -2;
    for (long i = 0; i < m; ++i) {
        const long j = i - 6;
        ext[i] = j < 0 ? 2.0 * x[0] - x[-j] : j >= n ? 2.0 * x[n - 1] - x[2 * (n - 1) - j] : x[j];
    }
    double z = zi * ext[0];
    for (long i = 0; i < m; ++i) { const double y = b0 * ext[i] + z; z = b1 * ext[i] - a1 * y; ext[i] = y; }
    z = zi * ext[m - 1];
    for (long i = m - 1; i >= 0; --i) { const double y = b0 * ext[i] + z; z = b1 * ext[i] - a1 * y; ext[i] = y; }
    for (long j = 0; j < n; ++j) out[j] = ext[j + 6];
    free(ext);
    return 0;
}

/* ---- Event.filter for any order (DataTypes.py:258-274 passes `order` straight to scipy.signal.bessel) ----------
 * scipy.signal.bessel(N, Wn, 'low', analog=False, output='ba') (norm='phase', the default), restated:
 *   analog prototype: the poles of the phase-normalised Bessel filter of order N (table below: what
 *     scipy.signal.besselap(N, 'phase') returns, 17 significant digits; conjugates implied), gain 1, no zeros;
 *   lp2lp: poles * wo, gain * wo^N, wo = 2 fs tan(pi Wn / fs) with fs = 2 (pre-warping);
 *   bilinear (fs = 2): p_z = (4 + p) / (4 - p), N zeros at -1, gain * Re(1 / prod(4 - p));
 *   b = gain * (z + 1)^N, a = prod(z - p_z)   (real coefficients).
 * scipy.signal.filtfilt(b, a, x): padlen = 3 (N + 1), odd extension, zi = lfilter_zi(b, a) -- the steady state of the
 * direct-form-II-transposed delays for a unit step: solve (I - A) zi = B with A[k][0] = -a[k+1], A[k][k+1] = 1,
 * B[k] = b[k+1] - a[k+1] b[0] --, forward lfilter from zi * ext[0], backward lfilter from zi * (last forward value).
 * Orders 1..8.  Returns 0; -1 for n <= padlen or a cutoff outside (0, Nyquist); -3 for an order outside 1..8. */
#include <complex.h>
#define SO_MAXORD 8
static const double so_bessel_poles[SO_MAXORD + 1][4][2] = {      /* [order][pair][re, im]; im == 0: a single real pole */
    {{0, 0}},
    {{-0.9999999999999998, 0.0}},
    {{-0.8660254037844384, 0.4999999999999999}},
    {{-0.9416000265332067, 0.0}, {-0.7456403858480766, 0.7113666249728351}},
    {{-0.9047587967882447, 0.27091873300387465}, {-0.6572111716718827, 0.830161435004873}},
    {{-0.9264420773877602, 0.0}, {-0.8515536193688396, 0.44271746394433265}, {-0.5905759446119191, 0.9072067564574549}},
    {{-0.9093906830472273, 0.1856964396793047}, {-0.7996541858328288, 0.5621717346937318}, {-0.5385526816693109, 0.9616876881954278}},
    {{-0.919487155649029, 0.0}, {-0.8800029341523375, 0.32166527623077396}, {-0.7527355434093214, 0.6504696305522552},
     {-0.4966917256672317, 1.0025085084544205}},
    {{-0.909683154665291, 0.1412437976671423}, {-0.8473250802359334, 0.42590175382729345}, {-0.7111381808485397, 0.7186517314108402},
     {-0.4621740412532123, 1.0343886811269012}},
};

int so_bessel_ba(int order, double wn, double *b, double *a)
{
    if (order < 1 || order > SO_MAXORD) return -3;
    if (!(wn > 0.0) || !(wn < 1.0)) return -1;
    const double wo = 4.0 * tan(3.14159265358979323846 * wn / 2.0);
    double complex p[SO_MAXORD];
    int np = 0;
    for (int k = 0; k < (order + 1) / 2; ++k) {
        const double re = so_bessel_poles[order][k][0], im = so_bessel_poles[order][k][1];
        if (im == 0.0) p[np++] = re * wo;
        else { p[np++] = (re + I * im) * wo; p[np++] = (re - I * im) * wo; }
    }
    double complex den = 1.0;
    for (int k = 0; k < order; ++k) den *= (4.0 - p[k]);
    const double gain = pow(wo, order) * creal(1.0 / den);
    double complex ac[SO_MAXORD + 1] = {1.0};                     /* prod (z - pz) */
    for (int k = 0; k < order; ++k) {
        const double complex pz = (4.0 + p[k]) / (4.0 - p[k]);
        for (int j = k + 1; j >= 1; --j) ac[j] = ac[j] - pz * ac[j - 1];
    }
    double bc[SO_MAXORD + 1] = {1.0};                             /* (z + 1)^order */
    for (int k = 0; k < order; ++k)
        for (int j = k + 1; j >= 1; --j) bc[j] = bc[j] + bc[j - 1];
    for (int j = 0; j <= order; ++j) { a[j] = creal(ac[j]); b[j] = gain * bc[j]; }
    return 0;
}

int so_lfilter_zi(int order, const double *b, const double *a, double *zi)
{
    double M[SO_MAXORD][SO_MAXORD + 1];                           /* (I - A | B), Gaussian elimination with pivoting */
    for (int r = 0; r < order; ++r) {
        for (int c = 0; c < order; ++c) M[r][c] = (r == c ? 1.0 : 0.0) - ((c == 0 ? -a[r + 1] : 0.0) + (c == r + 1 ? 1.0 : 0.0));
        M[r][order] = b[r + 1] - a[r + 1] * b[0];
    }
    for (int c = 0; c < order; ++c) {
        int piv = c;
        for (int r = c + 1; r < order; ++r) if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
        if (M[piv][c] == 0.0) return -1;
        for (int k = 0; k <= order; ++k) { const double t = M[c][k]; M[c][k] = M[piv][k]; M[piv][k] = t; }
        for (int r = 0; r < order; ++r) {
            if (r == c) continue;
            const double f = M[r][c] / M[c][c];
            for (int k = c; k <= order; ++k) M[r][k] -= f * M[c][k];
        }
    }
    for (int r = 0; r < order; ++r) zi[r] = M[r][order] / M[r][r];
    return 0;
}

static void so_lfilter(int order, const double *b, const double *a, double *v, long m, long step, const double *zi, double scale)
{
    double z[SO_MAXORD];
    for (int k = 0; k < order; ++k) z[k] = zi[k] * scale;
    double *q = step > 0 ? v : v + (m - 1);
    for (long i = 0; i < m; ++i, q += step) {
        const double x = *q, y = z[0] + b[0] * x;                 /* scipy's lfilter, in its operation order */
        for (int k = 0; k < order - 1; ++k) z[k] = (z[k + 1] + x * b[k + 1]) - y * a[k + 1];
        z[order - 1] = x * b[order] - y * a[order];
        *q = y;
    }
}

int so_bessel_filtfilt(const double *x, long n, int order, double cutoff, double second, double *out)
{
    double b[SO_MAXORD + 1], a[SO_MAXORD + 1], zi[SO_MAXORD];
    const int rc = so_bessel_ba(order, cutoff / (second / 2.0), b, a);
    if (rc) return rc;
    const long pad = 3 * (order + 1);
    if (n <= pad) return -1;
    if (so_lfilter_zi(order, b, a, zi)) return -2;
    const long m = n + 2 * pad;
    double *ext = (double *)malloc(sizeof(double) * (size_t)m);
    if (!ext) return -2;
    for (long i = 0; i < m; ++i) {
        const long j = i - pad;
        ext[i] = j < 0 ? 2.0 * x[0] - x[-j] : j >= n ? 2.0 * x[n - 1] - x[2 * (n - 1) - j] : x[j];
    }
    so_lfilter(order, b, a, ext, m, 1, zi, ext[0]);
    so_lfilter(order, b, a, ext, m, -1, zi, ext[m - 1]);
    for (long j = 0; j < n; ++j) out[j] = ext[j + pad];
    free(ext);
    return 0;
}

/* ==== calignment.pyx:20-100  cSegmentAligner =========================
 * TEST INFRASTRUCTURE like the rest of this file.  Pinned against the compiled, unmodified
 * reference by tests/golden/make_golden_align.py -> tests/golden/golden_align.npz.
 *
 * Semi-local alignment of a sequence of segments (mean, std, duration) to a model of
 * segments: score[i][j] = best score of aligning seq[0..i] with seq[i] on model[j]; moves
 * into (i, j): stay (i-1, j), step (i-1, j-1), skip forward over model segments (running
 * maximum from the left, skip_penalty x skipped duration), slip back (running maximum
 * from the right, backslip_penalty x duration).
 *
 * Return codes follow what the compiled reference does on the same input:
 *    0  ok
 *    1  ValueError   (s == 0: the reference cannot create an empty (0, m) array)
 *    2  IndexError   (an index left [0, m): m == 1 with s > 1; the traceback reaching j == 0
 *                     before the first segment -- `score[i-1, j-1]` with an unsigned j; a skip to
 *                     before the model start)
 *    3  ZeroDivisionError (seq_std * model_std == 0)
 *    4  no final score above -1: double_argmax (calignment.pyx:11-18) starts from -1 and
 *       returns an UNINITIALISED int then -- undefined in the reference, reported here
 * out_score = score[s-1][m-1]  (the reference divides it by numpy's sum of the durations; callers do
 * that in numpy so that the pairwise summation order is numpy's). out_path: s entries (uint32: the
 * reference keeps j unsigned, a final skip below 0 shows as 4294967295).                          */
#define SO_NEGINF (-99999999.0)
static double so_dmax(double a, double b) { return a >= b ? a : b; }    /* :8 */

int so_align(const double *mm, const double *ms, const double *md, int m,
             double skip_pen, double back_pen,
             const double *sm, const double *ss, const double *sd, int s,
             double *out_score, unsigned *out_path)
{
    if (s <= 0) return 1;
    if (m <= 0) return 1;
    size_t cells = (size_t)s * (size_t)m;
    double *match = malloc(cells * sizeof(double)), *score = malloc(cells * sizeof(double));
    double *skip = malloc(cells * sizeof(double)), *back = malloc(cells * sizeof(double));
    double *cdur = malloc((size_t)m * sizeof(double));
    int rc = 0;
#define AT(a, i, j) a[(size_t)(i) * (size_t)m + (size_t)(j)]
    { double run = 0.0;                                     /* :30 np.cumsum: sequential */
      for (int j = 0; j < m; ++j) { run = j ? run + md[j] : md[j]; cdur[j] = run; } }
    for (int i = 0; i < s && !rc; ++i)                      /* :47-49 */
        for (int j = 0; j < m; ++j) {
            double d = sm[i] - mm[j], den = ss[i] * ms[j];
            if (den == 0.0) { rc = 3; break; }
            AT(match, i, j) = -(d * d) / den;
        }
    if (rc) goto done;
    for (int j = 0; j < m; ++j)                             /* :51-52 */
        AT(score, 0, j) = AT(match, 0, j) * sd[0] - skip_pen * (cdur[j] - md[j]);
    if (s > 1 && m < 2) { rc = 2; goto done; }              /* :59-62 index 1 / m-2 of a 1-wide model */
    for (int i = 1; i < s; ++i) {
        AT(skip, i, 0) = SO_NEGINF;                         /* :55-57 */
        for (int j = 1; j < m; ++j)
            AT(skip, i, j) = so_dmax(AT(skip, i, j - 1), AT(score, i - 1, j - 1)) - md[j] * skip_pen;
        AT(back, i, m - 1) = SO_NEGINF;                     /* :58-61 (row j = 0 has the same form) */
        for (int j = m - 2; j >= 0; --j)
            AT(back, i, j) = so_dmax(AT(back, i, j + 1), AT(score, i - 1, j + 1)) - md[j + 1] * back_pen;
        for (int j = 0; j < m; ++j) {                       /* :63-69 */
            double p = AT(score, i - 1, j);
            if (j > 0) {
                if (AT(score, i - 1, j - 1) > p) p = AT(score, i - 1, j - 1);   /* Python max(): first of equals */
                if (AT(skip, i, j - 1) > p) p = AT(skip, i, j - 1);
            }
            if (j < m - 1) p = so_dmax(p, AT(back, i, j));
            AT(score, i, j) = p + AT(match, i, j) * sd[i];
        }
    }
    {
        unsigned j = 0, mu = (unsigned)m;
        double best = -1.0; int found = 0;                  /* :11-18 */
        for (int q = 0; q < m; ++q)
            if (AT(score, s - 1, q) > best) { best = AT(score, s - 1, q); j = (unsigned)q; found = 1; }
        if (!found) { rc = 4; goto done; }
        for (int i = s - 1; i >= 1; --i) {                  /* :73-97 */
            out_path[i] = j;
            if (j >= mu || j == 0) { rc = 2; goto done; }   /* score[i, j] / score[i-1, j-1] out of bounds */
            double prev = AT(score, i, j) - AT(match, i, j) * sd[i];
            double tol = 1e-6 * fabs(prev);
            if (fabs(prev - AT(score, i - 1, j - 1)) <= tol) { j -= 1; continue; }
            if (fabs(prev - AT(score, i - 1, j)) <= tol) continue;
            if (j < mu - 1) {
                unsigned k = j; double t = prev;
                while (k < mu - 1 && fabs(t - AT(back, i, k)) <= tol) { k += 1; t += md[k] * back_pen; }
                if (k > j) { j = k; continue; }
            }
            if (j > 0) {
                unsigned k = j; double t = prev;
                while (k >= 1 && fabs(t - AT(skip, i, k - 1)) <= tol) { k -= 1; t += md[k] * skip_pen; }
                if (k < j) { j = k - 1u; continue; }
            }
        }
        out_path[0] = j;
        *out_score = AT(score, s - 1, m - 1);
    }
done:
#undef AT
    free(match); free(score); free(skip); free(back); free(cdur);
    return rc;
}
