/*
 * poreseg.h -- C ABI of libporeseg.so, the MI355X (gfx950) implementation of PyPore's
 * SpeedyStatSplit / FastStatSplit change-point segmenter.
 *
 * Every entry point below replaces one piece of the reference interface (file:line under
 * the reference tree, PyPore/...).  The reference has no FFI of its own (the hot path is a
 * Cython cdef class called from Python); the binding a maintainer would add is a ctypes
 * stub, shown in INTEGRATION.md, which is exactly what pypore_amd/_lib.py does.
 *
 * Conventions: plain C, no exceptions cross the boundary.  Every function returns a status
 * (PS_OK == 0, negative on error); ps_last_error() returns a human-readable message for the
 * most recent failure on that context.  Pointers prefixed d_ are DEVICE pointers (HBM of the
 * context's GPU), h_ are host pointers.  Outputs are caller-allocated with an explicit
 * capacity; when a capacity is too small the call fails with PS_ERR_CAPACITY and reports
 * the required size through the corresponding out-parameter.  A context is bound to one
 * GPU and one HIP stream and may be used by one host thread at a time.
 *
 * Sample model.  The reference consumes float64 current in pA (cparsers.pyx:53,103).  Real
 * traces are int16 ADC counts times a scale (read_abf.py:202-210), so the device consumes
 * either fp32 pA values that lie on an ADC grid (x = k * quantum, k integer, |k| < 2^23) or
 * the raw int16 counts; prefix sums are then exact integers and window variances are
 * evaluated in fp64 with the reference's operation order.  Off-grid fp32 input is rejected
 * with PS_ERR_OFF_GRID (never silently rounded).
 */
#ifndef PORESEG_H
#define PORESEG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PS_OK                 0
#define PS_ERR_ARG           -1   /* invalid argument (null pointer, negative size ...) */
#define PS_ERR_ASSERT_WIDTH  -2   /* reference assertion max_width >= min_width        (cparsers.pyx:69) */
#define PS_ERR_ASSERT_WINDOW -3   /* reference assertion window_width >= 2*min_width   (cparsers.pyx:71) */
#define PS_ERR_ASSERT_CUTOFF -4   /* reference assertion cutoff_freq <= sampling_freq/2 (cparsers.pyx:74) */
#define PS_ERR_CAPACITY      -5   /* caller buffer too small; required size reported */
#define PS_ERR_OFF_GRID      -6   /* fp32 sample is not an integer multiple of quantum */
#define PS_ERR_HIP           -7   /* HIP runtime error (message in ps_last_error) */
#define PS_ERR_NO_DEVICE     -8   /* no gfx950 device / code object missing */
#define PS_ERR_INTERNAL      -9   /* device-side stack or scratch overflow */

#define PS_DTYPE_F32 0            /* float pA on the grid k*quantum */
#define PS_DTYPE_I16 1            /* raw int16 ADC counts (read_abf.py:208) */
#define PS_DTYPE_F64 2   /* float64 pA on no grid: accepted by ps_filter_bessel ONLY (the current of an event that was
                            filtered before, DataTypes.py:258-274); quantum and offset_counts are ignored */

typedef struct ps_ctx ps_ctx;

/* The eight constructor arguments of FastStatSplit / SpeedyStatSplit
 * (cparsers.pyx:55-57, parsers.py:511-513).  "Not given" (Python None) is encoded as 0,
 * which the reference also treats as not given (`if not false_positive_rate`). */
typedef struct ps_split_params {
    int32_t min_width;                   /* default 100      */
    int32_t max_width;                   /* default 1000000  */
    int32_t window_width;                /* default 10000    */
    double  min_gain_per_sample;         /* 0 = None         */
    double  false_positive_rate;         /* 0 = None         */
    double  prior_segments_per_second;   /* 0 = None         */
    double  sampling_freq;               /* default 1e5      */
    double  cutoff_freq;                 /* 0 = None         */
} ps_split_params;

/* How samples map to pA: pA = (count + offset_counts) * quantum for PS_DTYPE_I16; PS_DTYPE_F32 samples are the pA
 * values themselves, count = x/quantum, and offset_counts is NOT added to them: non-zero, it names the level (in counts
 * of quantum) that the caller subtracted upstream -- ps_requantise's centre, for a filtered event -- and is used only to
 * judge near ties against the reference's own rounding noise (its cumsums run on the uncentred values; counters[11] of
 * ps_get_timings).  quantum should be a power of two for bit-exact parity with the reference (read_abf.py:202-205
 * scale/offset). */
typedef struct ps_sample_format {
    int32_t dtype;                       /* PS_DTYPE_* */
    int32_t offset_counts;
    double  quantum;
} ps_sample_format;

/* Per-segment statistics: Segment.mean/std/min/max (core.py:209-223), std = population. */
typedef struct ps_segstat {
    double mean, std, min, max;
} ps_segstat;

/* Library / device ------------------------------------------------------------------------ */
const char *ps_version(void);
/* Number of visible gfx950 GPUs (0 if none; never fails). */
int ps_device_count(void);
/* Creates a context on GPU `device`; stream = an existing hipStream_t to launch on, or NULL
 * to let the context create its own. */
int ps_create(int device, void *stream, ps_ctx **out);
void ps_destroy(ps_ctx *ctx);
const char *ps_last_error(const ps_ctx *ctx);
/* Tiling of long traces: a trace longer than tile_len + halo samples is cut into tiles whose
 * spines are computed speculatively and stitched.  0 keeps the default. */
int ps_set_tiling(ps_ctx *ctx, int64_t tile_len, int64_t halo);
/* Diagnostic / tuning options (none changes a result): "mode" 0 screen+exact (default), 1 exact fp64
 * scans only, 2 verify (screen and exact must agree); "scan_bs" 1 (default) block-sum scan with
 * single-wave workgroups behind the block-prefix kernel, 0 LDS-window scan; "prune" 0 switches the
 * block pruning of the LDS-window scan off; "stitch_host" 1 forces the host-stitch pipeline (halo
 * tiles + seam repairs, otherwise only the fallback); "timing" 0/1/2 (see ps_get_timings); "upload_by_kernel" 1
 * (default) the call's host tables are fetched by a kernel reading the pinned blob, 0 hipMemcpyAsync; "filter_fused" 1
 * (default) fast filters run both directions in one kernel over tiles with halos, 0 always the exact three-pass
 * scan; "tree_mw" 0 (default) subtree jobs of the block-sum scan run on single-wave workgroups, one per wave slot,
 * striding over the job list, 1 two-wave workgroups whose waves share the workgroup's job list through an LDS counter
 * (round 2's default; slower at four waves per SIMD); "groups" 1 (default) K0 writes one record per 256 samples and
 * every window scan starts with the coarse pass over them (whole groups of 32 blocks are bounded, rows of the sweep
 * that lie in pruned groups are skipped), 0 every row is swept; "wide_bs" 1 (default) counts too wide for the 32-bit
 * digest are retried on the 64-bit one, 0 straight to the LDS-window scan; "spine_nt" 256/512/1024, "tree_nt" 256/512
 * workgroup sizes of the LDS-window kernels; "tree_par" 1 (default) the deep subtree jobs of a call on the 64-bit digest
 * (a filtered event) are shared by the four waves of a workgroup when the call has few jobs, 0 one wave per job;
 * "k0_waves" n > 0: the block-prefix kernel K0 is persistent -- n waves per SIMD (twice that for int16 samples) stride
 * over the call, each with its next wave block's samples in flight: what a context that shares the chip with other calls
 * wants (engine.StreamPool sets 1) --, 0 (default): one wave per 2 048 samples, as many as fit the chip: faster for a
 * call that has the chip to itself;
 * "k0_shared" 1: this context queues its upload + K0 launches on the device's shared front stream, where the K0
 * kernels of all such contexts run back to back (they are bound by HBM: side by side they only share it) and the
 * context's own stream takes over behind an event; 0 (default) everything on the context's stream -- a host that keeps
 * several contexts busy (engine.StreamPool) sets it for the duration of its runs (measured slower: nothing sets it);
 * "k0_admit" M > 0: at most M calls of this device have their K0 in flight at a time -- enforced on the device (round 6): a
 * call queues a wait for the K0 of the call M tickets before it in front of its own K0 (hipStreamWaitEvent); the host thread
 * never waits --, 0 (default) no limit; engine.StreamPool sets 3 ("shared_device");
 * "bridge_ext" 1 (default): seams whose bridge ran out of anchors (256 without meeting a downstream tile's chain:
 * densely stepped data) and open tiles entered exactly at their start are continued on the device -- up to four rounds,
 * 16 384 more anchors per seam -- before the call falls back to the host stitch, 0 straight to the host stitch;
 * "lat_help" 1 (default): in the look-ahead kernel, workgroups that are through with their own seams scan chunks of 16
 * windows ahead of the seams that walk long stretches without splits and publish what they find; the seam's owner takes a
 * published chunk instead of scanning it (2 x on traces with stretches of 1e6 samples and more), 0 every seam walks alone,
 * 2 as 1 but the helpers stay until every workgroup of the launch is through with its own seams (never leave on idle polls:
 * for tests that must see them work -- counters[12], [13] of ps_get_timings; only for a call that has the chip to itself);
 * "bridge_budget" 1..256 (default 256): anchors a bridge may add before it gives up (tests lower it to reach the second
 * chance on small inputs);
 * "shared_device" n: ONE call for a host that keeps n contexts busy on one device (a pool of host threads, one context
 * each): n > 1 sets k0_waves 1, lat_help 0 and, for n > 3, k0_admit 3 -- the measured settings of a shared chip
 * (INTEGRATION.md has the table) --, n <= 1 restores the defaults of a lone context (k0_waves 0, lat_help 1, k0_admit 0);
 * "single_pass" 1 (default): ps_detect_segment_trace streams a file trace once (see there), 0: it makes the two calls;
 * "gather_fused" 1 (default): the gather places its items from per-256-job count sums (no scan kernel in front of it), 0: rounds
 * 2-5's item scan + gather; "download_by_kernel" 1 (default): the status block returns by a kernel writing pinned memory;
 * "k0_unaligned" 0: K0's fast route only from 16-byte-aligned addresses (1: whatever the probe at ps_create said);
 * "debug" 1: the library reports on stderr which occupancy it found and which seams gave up (prints only);
 * "slots_pct" 1..100 (default 100) share of the resident wave slots the single-wave scan kernels are launched on;
 * "tree_jobs_per_wave" (default 4) subtree kernel: jobs / this many of its slots work, between half and all of them;
 * "noise_k_ppm" (default 100 000 = 0.1): near-tie accounting of the 64-bit digest, margin factor in millionths.
 * Unknown names return PS_ERR_ARG.
 * THE LIBRARY READS NO ENVIRONMENT VARIABLE (round 6): what a call returns depends on its arguments and on these two
 * setters only.  The experiments' switches that return stale or partial results ("dbg_phase", "dbg_k0_nogrp",
 * "scan_lds_pad", "rep_*") and the PORESEG_* variables exist in libporeseg_diag.so only (make -C pypore_amd/csrc diag;
 * ps_version() of that build says DIAGNOSTIC BUILD). */
int ps_set_option(ps_ctx *ctx, const char *name, int64_t value);
/* Blocks until all work submitted on the context's stream has finished. */
int ps_synchronize(ps_ctx *ctx);

/* Replaces FastStatSplit.__init__ (cparsers.pyx:55-101): validates the widths with the
 * reference's three assertions and computes the public attribute `min_gain`. Host only. */
int ps_min_gain(const ps_split_params *p, double *min_gain_out);

/* Replaces FastStatSplit.parse (cparsers.pyx:103-118 -> _recursive_split :180-203 ->
 * _best_split_stepwise :157-178 -> var_c :31-38) for a batch of independent events
 * (one reference parse() call per event; Event.parse, DataTypes.py:286).
 *   d_samples  all events' samples, one contiguous device array
 *   h_ev_off   n_ev+1 host offsets into d_samples (event e = [h_ev_off[e], h_ev_off[e+1]))
 *   d_bounds   device out, capacity `cap` int32: breakpoints of event 0, then event 1, ...
 *              (event-local sample indices, ascending, excluding 0 and the event length --
 *              the list _recursive_split returns)
 *   h_bounds_off  host out, n_ev+1 offsets into d_bounds
 *   d_stats    nullable device out, capacity `cap + n_ev` entries: statistics of the
 *              segments of event 0, then event 1, ... (segment s of event e lives at
 *              h_bounds_off[e] + e + s)
 * Returns PS_ERR_CAPACITY if cap < total (h_bounds_off[n_ev] then holds the required size). */
int ps_segment_batch(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt,
                     const int64_t *h_ev_off, int32_t n_ev,
                     const ps_split_params *params,
                     int32_t *d_bounds, int64_t cap, int64_t *h_bounds_off,
                     ps_segstat *d_stats);

/* ps_segment_batch plus one more output: d_is_spine (nullable device out, capacity `cap` bytes) is 1
 * for breakpoints found by the top-level chain of right recursions rec(a, len) ("spine anchors"),
 * 0 for breakpoints of left subtrees.  Two runs over overlapping pieces of one trace are identical
 * after any common spine anchor; pypore_amd/dist.py uses that to stitch a trace sharded across
 * GPUs (BASELINE config 5).  No reference counterpart. */
int ps_segment_batch_ex(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt,
                        const int64_t *h_ev_off, int32_t n_ev,
                        const ps_split_params *params,
                        int32_t *d_bounds, int64_t cap, int64_t *h_bounds_off,
                        ps_segstat *d_stats, uint8_t *d_is_spine);

/* Same as ps_segment_batch_ex for events that do NOT tile the sample array: event e is
 * [h_ev_start[e], h_ev_start[e] + h_ev_len[e]) of d_samples (the events lambda_event_parser cuts out
 * of a file trace, parsers.py:142-155 -> Event.parse, DataTypes.py:978-984, without copying them). */
int ps_segment_events(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt,
                      const int64_t *h_ev_start, const int64_t *h_ev_len, int32_t n_ev,
                      const ps_split_params *params,
                      int32_t *d_bounds, int64_t cap, int64_t *h_bounds_off,
                      ps_segstat *d_stats, uint8_t *d_is_spine);

/* The EXACT route for float64 input that lies on no ADC grid (round 6).  The reference takes any double[:] (cparsers.pyx:53)
 * and its decisions on such data depend on the rounding of its own prefix sums, c = np.cumsum(current) and c2 =
 * np.cumsum(np.multiply(current, current)) (cparsers.pyx:110-111): strictly sequential fp64 additions.  This entry forms
 * exactly those -- one chain per event, the events in parallel -- and scans every window of the recursion
 * (cparsers.pyx:157-203) with the reference's own expressions on them (var_c, :31-38; gain = var_summed - (low + high);
 * strict '>', first maximum): the same boundaries as the reference on the same input BY CONSTRUCTION, up to the logarithm's
 * last bit (counters[11] counts nothing here; a tie within 1e-12 of a gain is as undecided as everywhere else).
 *   d_current   float64 pA, device; event e = [h_ev_start[e], h_ev_start[e] + h_ev_len[e])
 *   d_bounds / cap / h_bounds_off   as ps_segment_batch; no statistics (the caller has the float64 values)
 * Cost: ~10-20 ns per sample and event for the chains (events run side by side), then one 512-thread workgroup per window
 * with two fp64 logarithms per candidate: a 1e6-sample event takes tens of milliseconds where the re-quantised route
 * (ps_requantise + ps_segment_batch) takes one.  Meant for the events that route flags (near ties) and for callers who
 * want the reference's own arithmetic; pypore_amd: SpeedyStatSplit(off_grid="exact" / "exact_on_near_tie"). */
int ps_segment_exact_f64(ps_ctx *ctx, const double *d_current, const int64_t *h_ev_start, const int64_t *h_ev_len,
                         int32_t n_ev, const ps_split_params *params, int32_t *d_bounds, int64_t cap,
                         int64_t *h_bounds_off);

/* Upper bound on the number of breakpoints ps_segment_batch can emit for these events
 * (sum over events of len/min_width): a safe `cap`. */
int64_t ps_bounds_capacity(const int64_t *h_ev_off, int32_t n_ev, int32_t min_width);

/* Replaces FastStatSplit.best_single_split (cparsers.pyx:120-155): one scan of the whole
 * array, start=0, end=n-1, candidates range(2, end-2), threshold 0.  Returns (gain, index)
 * or (0.0, -1). */
int ps_best_single_split(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt,
                         int64_t n, double *gain_out, int32_t *index_out);

/* Replaces FastStatSplit.score_samples(current, no_split=True) (cparsers.pyx:205-249):
 * per-candidate gains of ONE window spanning the whole array; d_scores[n] (device) gets the
 * gain at every candidate index and 0 elsewhere; *split_out the chosen split or -1. */
int ps_score_window(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt,
                    int64_t n, int32_t min_width, double min_gain,
                    double *d_scores, int32_t *split_out);

/* Diagnostic (tests only; no reference counterpart): audits the pruning bounds of the block-sum window scan ON THE
 * DEVICE, with the scan code of the product.  The trace [0, n) is taken as one event (K0 digest), every window
 * [h_windows[2i], h_windows[2i+1]) is scanned with all rows swept, and each bound the scan forms is compared with the
 * screened gains of the candidates it covers, evaluated one by one from the raw samples:
 *   out[0] blocks with a corner bound        out[1] of which violated      out[6] smallest margin (bound - largest gain)
 *   out[2] blocks with a two-boundary bound  out[3] of which violated      out[7] smallest margin
 *   out[4] groups (256 samples) with a bound out[5] of which violated      out[8] smallest margin
 *   out[9] windows with a coarse pass
 * A violation is a covered gain above the bound by more than 2 delta(n), the slack every pruning level carries
 * (DESIGN.md 4.3, 4.4).  Needs min_width >= 8 and counts inside the 32-bit digest (PS_ERR_ARG otherwise). */
int ps_audit_bounds(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt, int64_t n,
                    const ps_split_params *params, const int32_t *h_windows, int32_t n_win, double *out12);

/* Replaces lambda_event_parser(threshold).parse with the default rules (parsers.py:124-155, rules
 * :133-135): events are the maximal runs of samples on one side of `threshold` (mask = x < threshold,
 * cut at every mask edge) that satisfy  length > min_duration,  min > min_current,  max < threshold.
 * Reference defaults: threshold 90, min_duration 100000, min_current -0.5.  Writes the kept events'
 * (start, length) in samples, ascending, to the host arrays; *n_events_out is the number found
 * (PS_ERR_CAPACITY if it exceeds cap).  One streaming pass over the trace (edge positions + per-chunk
 * min/max), then one workgroup per long piece. */
int ps_detect_events(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt, int64_t n,
                     double threshold, int64_t min_duration, double min_current,
                     int64_t *h_starts, int64_t *h_lengths, int64_t cap, int64_t *n_events_out);

/* File.parse + Event.parse for every event of a whole file trace in ONE call and ONE pass over its samples (round 6):
 * ps_detect_events followed by ps_segment_events on the events it found, with the same results -- events (h_starts,
 * h_lengths, *n_events_out as ps_detect_events), boundaries (d_bounds / cap / h_bounds_off[n_events + 1] as ps_segment_events;
 * h_bounds_off must have room for ev_cap + 1 entries), optional statistics.  The block-sum kernel K0 runs once over the whole
 * trace (blocks aligned to the trace) and judges every 8-sample block against the threshold on its way (2 bits per block: all
 * below, all at or above, mixed; min / max per 1 024 samples); the detector reads those bits -- 1/64 of the samples' bytes --
 * and the samples of the mixed blocks only; the events -- which start at any sample -- are segmented from the same digest in
 * coordinates shifted by (start mod 8).  Replaces the loop `for event in file.parse(lambda_event_parser(threshold)): event.parse(SpeedyStatSplit(...))`
 * (parsers.py:124-155, DataTypes.py:589-602, 978-984).  Takes the two calls by itself when the block-sum scan does not apply
 * (min_width < 8, window_width > 64 512, options "scan_bs" 0 / "mode" 1 / "stitch_host" 1 / "single_pass" 0) or when a count of
 * the trace lies 2^14 or more from its first one (ps_get_timings counters[7] = 3 then). */
int ps_detect_segment_trace(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt, int64_t n,
                            double threshold, int64_t min_duration, double min_current,
                            const ps_split_params *params,
                            int64_t *h_starts, int64_t *h_lengths, int64_t ev_cap, int64_t *n_events_out,
                            int32_t *d_bounds, int64_t cap, int64_t *h_bounds_off, ps_segstat *d_stats);

/* Replaces Event.filter (DataTypes.py:258-274): scipy.signal.bessel(order, cutoff / (sampling_freq / 2), btype='low',
 * analog=0) applied with scipy.signal.filtfilt (forward and backward, odd extension by padlen = 3 (order + 1), initial
 * state lfilter_zi * first value).  order 1 (the reference's default): the scan / fused-halo kernels; orders 2..8: one
 * thread per segment with a halo (seg_filter.hpp).  Input as for the segmenter (fp32 on the grid or int16 counts,
 * n samples) or -- this entry only -- PS_DTYPE_F64, the float64 current itself; output d_out[n] in pA as fp64 (the
 * reference replaces Event.current by the float64 result).
 * PS_ERR_ARG for order outside 1..8, n <= padlen (scipy raises ValueError), a cutoff outside (0, Nyquist), and for a
 * filter of order >= 2 whose state needs more than 8192 samples to forget (cutoffs below ~0.2 % of Nyquist). */
int ps_filter_bessel(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt, int64_t n, int32_t order,
                     double cutoff, double sampling_freq, double *d_out);

/* The representation step between ps_filter_bessel and ps_segment_batch for a filtered event (no reference counterpart:
 * the reference segments the float64 result of Event.filter directly, DataTypes.py:286; this library works on exact
 * integer sums).  d_in[n]: a filtered current in pA (fp64, device).  Writes d_out[n] (fp32, device): the current minus
 * *centre_out, rounded to multiples of *step_out -- centre = the mean rounded to the grid, step = the finest power of two
 * that keeps every count below 2^22 (what DataTypes.Event.parse does on the host for a single event).  Segment d_out with
 * quantum = *step_out; the gains are shift invariant, so the boundaries are those of the rounded current.  Synchronises
 * the context's stream twice: for the statistics (they come back to the host) and before returning (d_out is complete
 * and d_in no longer read when the call returns). */
int ps_requantise(ps_ctx *ctx, const double *d_in, int64_t n, float *d_out, double *centre_out, double *step_out);

/* Event.filter + the representation step for MANY events of one trace in one call (the inner loop of Experiment.parse,
 * DataTypes.py:975-984; File.parse_events here): event e is samples[ev_start[e] .. + ev_len[e]) of d_samples; its filtered
 * current (ps_filter_bessel) goes to d_filtered + sum of the lengths before it, its re-quantised copy (ps_requantise) to the
 * same position of d_rounded, its centre and grid step to h_centre[e] / h_step[e].  Same results as the two entries called
 * per event; one status check and two host synchronisations for the whole batch instead of three per event. */
int ps_filter_requantise_batch(ps_ctx *ctx, const void *d_samples, const ps_sample_format *fmt, const int64_t *ev_start,
                               const int64_t *ev_len, int32_t n_ev, int32_t order, double cutoff, double sampling_freq,
                               double *d_filtered, float *d_rounded, double *h_centre, double *h_step);

/* Replaces cSegmentAligner(model_means, model_stds, model_durs, skip_penalty, backslip_penalty).align(seq_means,
 * seq_stds, seq_durs) (calignment.pyx:20-100; called from SegmentAligner.align, alignment.py:33-46) for a BATCH of
 * sequences against one model: sequence q is entries [h_seq_off[q], h_seq_off[q+1]) of the three device arrays
 * (per-segment mean, std, duration -- e.g. the statistics ps_segment_batch produced).  Model arrays are host
 * pointers (m doubles each, 1 <= m <= 1024).  Outputs (device): d_scores[q] = score[s-1][m-1] -- the reference
 * returns it divided by numpy.sum(seq_durs), which the caller does in numpy to keep numpy's summation order --,
 * d_paths = the model index of every sequence segment (uint32 like the reference's unsigned j; same layout as the
 * inputs), d_status[q] = what the compiled reference does on that sequence:
 *   PS_ALIGN_OK; PS_ALIGN_VALUE_ERROR (empty sequence); PS_ALIGN_INDEX_ERROR (one-segment model with s > 1, or the
 *   traceback reaching model index 0 before the first sequence segment: `score[i-1, j-1]`, calignment.pyx:76);
 *   PS_ALIGN_ZERO_DIVISION (seq_std * model_std == 0, :49); PS_ALIGN_UNDEFINED (no final score above -1:
 *   double_argmax, :11-18, returns an uninitialised int there).  Scores and paths are bit-exact with the reference:
 *   the kernel keeps its operation order (sequential running maxima, fp64, no FMA contraction). */
#define PS_ALIGN_OK             0
#define PS_ALIGN_VALUE_ERROR    1
#define PS_ALIGN_INDEX_ERROR    2
#define PS_ALIGN_ZERO_DIVISION  3
#define PS_ALIGN_UNDEFINED      4
int ps_align_batch(ps_ctx *ctx, const double *h_model_means, const double *h_model_stds, const double *h_model_durs,
                   int32_t m, double skip_penalty, double backslip_penalty, const double *d_seq_means,
                   const double *d_seq_stds, const double *d_seq_durs, const int64_t *h_seq_off, int32_t n_seq,
                   double *d_scores, uint32_t *d_paths, int32_t *d_status);

/* Timing of the most recent ps_segment_batch, measured with HIP events on the context's stream.
 * ms[7] = the call's device work from the first upload to the last result copy (option "timing" >= 1, the
 * default); ms[3] = whole call on the host's wall clock.  With option "timing" = 2 an event is also recorded
 * between the phases: ms[0] spine kernel, ms[1] tree kernel, ms[2] gather+stats kernels, ms[4] stitch (assemble
 * kernels, or the host stitch), ms[5] bridge kernels, ms[6] block-prefix kernel (K0; 0 for the LDS-window scan)
 * -- every such event keeps the next kernel from starting back to back (about 6 us of idle GPU each), so the
 * breakdown is a diagnostic and off by default.  counters[0] window scans, [1] candidate
 * positions covered, [2] tiles, [3] tree jobs, [4] seams mended: 1 .. 999 999 seams continued on the device (option
 * "bridge_ext"), 1 000 000 + n the call was redone by the host stitch, with n repairs, [5] windows decided in fp64 (among
 * contenders or by a whole-window scan), [6] of which whole-window scans, [7] 1 = the call was redone on the 64-bit
 * digest, 2 = on the LDS-window kernels (counts too wide for the block sums), [8] [9] [10] window scans of the spine /
 * bridge / subtree kernels, [11] near ties: windows decided among fp64 contenders whose margin -- winner against the best
 * other candidate, or against min_gain -- is below 1e-9 * max(1, |gain|).  The device logarithm is not glibc's bit for bit
 * (gains differ by ~1e-11), so the reference could have decided such a window the other way; exact ties are decided like
 * the reference (first maximum wins, cparsers.pyx:175-177) and counted too; -1 = not counted: the call ran on the
 * LDS-window kernels (min_width < 8, option "scan_bs" 0 / "stitch_host" 1, counters[7] = 2, ps_segment_exact_f64); [12] chunk results (16 windows each) that the
 * look-ahead kernel's helpers published (option "lat_help"), [13] published chunks that a seam's owner took instead of
 * scanning them. */
int ps_get_timings(const ps_ctx *ctx, double *ms, int32_t n_ms, int64_t *counters, int32_t n_counters);
/* The fourteen work counters of ps_get_timings in place (valid for the life of the context; host memory, updated by every
 * call before it returns): a caller that checks one of them after each call -- counters[11], the near ties -- reads it
 * there instead of making a second call. */
const int64_t *ps_counters(const ps_ctx *ctx);

/* Synthetic step-signal generator (bit-identical to pypore_amd/synth.py):
 * sample i = level_counts[segment containing i] + noise(seed, i), written as fp32 pA
 * (count * 2^-5) or int16 counts.  h_seg_end[nseg] are the exclusive segment ends
 * (ascending, last >= n).  Bench/test infrastructure. */
int ps_synth_trace(ps_ctx *ctx, void *d_out, int32_t dtype, int64_t n, uint64_t seed,
                   const int64_t *h_seg_end, const int32_t *h_level_counts, int64_t nseg);

#ifdef __cplusplus
}
#endif
#endif /* PORESEG_H */
