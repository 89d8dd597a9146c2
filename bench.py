#!/usr/bin/env python3
"""bench.py -- Msamples/s segmented by SpeedyStatSplit on a 10^8-sample trace (BASELINE.json).

One "step" = one pass of the hot path (ps_segment_batch: block-prefix kernel K0 -> spine ->
bridge -> stitch -> tree -> gather) over one 10^8-sample synthetic trace that is already resident in HBM.  With
--gpus N every rank segments its own trace (weak scaling, no data-path collective; one RCCL
all_gather of the boundary counts after the timed region).

Prints ONE JSON line (rank 0).  `roofline` prices the kernel sequence of one step (sum of the
launch durations, HIP events on the library's stream) at 4 B per input sample against 8 TB/s, and
lists every kernel with its own algorithmic bytes and measured HBM traffic; `cpu_baseline` times
the CPU oracle (oracle/, a port of the reference) on a bounded prefix of the same trace on this host.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PARAMS = dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10.,
              sampling_freq=1e5)
HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md
BYTES_PER_SAMPLE = 4       # one fp32 read per input sample
# HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (r01_bs_pmc_summary.txt):
# FETCH_SIZE (KiB) x2 per the gfx950 correction + WRITE_SIZE (KiB), fp32 trace workload, default build.
PMC_TRAFFIC = {"blocksum_ms": (2 * 195356 + 196838) * 1024, "spine_ms": (2 * 186039 + 310) * 1024,
               "bridge_ms": (2 * (3940 + 155) + 86) * 1024, "tree_ms": (2 * 87959 + 1428) * 1024}
PMC_TRAFFIC_BYTES = sum(PMC_TRAFFIC.values())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--samples", type=int, default=100_000_000)
    ap.add_argument("--cpu-samples", type=int, default=20_000_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-detail", action="store_true", help="skip the separate per-kernel timing pass (kernel_ms stays 0)")
    ap.add_argument("--stats", action="store_true", help="also run the per-segment statistics kernel in the step")
    ap.add_argument("--dwell", type=int, nargs=2, default=None, metavar=("LO", "HI"),
                    help="dwell range of the synthetic trace in samples (default: BASELINE's U[1000, 20000))")
    ap.add_argument("--workload", choices=["trace", "file"], default="trace",
                    help="trace: one SpeedyStatSplit.parse over the whole 1e8-sample fp32 trace (default); "
                         "file: BASELINE config 3 -- int16 .abf-shaped trace, lambda_event_parser(threshold=90) "
                         "then per-event SpeedyStatSplit, end to end on the GPU")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pypore_amd import _lib, engine, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = args.gpus > 1 or world > 1 or os.environ.get("PORESEG_BENCH_DIST") == "1"   # env: exercise the
    if use_dist:                                         # N > 1 code path on one GPU (torchrun --nproc-per-node 1)
        assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.cuda.current_device()
    ctx = engine.context(dev)

    n = args.samples
    seed = 2024 + rank                                   # rank 0's trace is golden case G7
    params = _lib.split_params(**PARAMS)
    from pypore_amd import dist as pdist
    if args.workload == "trace":
        d = synth.dwell_table(seed, n, *args.dwell) if args.dwell else synth.dwell_table(seed, n)
        ends = np.cumsum(d)
        lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
        trace = ctx.synth_trace(n, seed, ends, lv, dtype=torch.float32)
        ev_off = np.array([0, n], dtype=np.int64)
        out1 = torch.empty(n // PARAMS["min_width"] + 1, dtype=torch.int32, device=trace.device)   # reused result buffer

        def step(k=None):
            # k: index of a timed batch when the job gathers its boundaries at the end (N > 1): they are written straight
            # into row k of the send buffer
            b, o, st = ctx.segment_batch(trace, ev_off, params, synth.QUANTUM, want_stats=args.stats,
                                          out=acc[k] if k is not None and acc is not None else out1)
            if k is not None and acc is not None:
                acc_counts[k] = b.numel()
            return b, o, st
    else:
        ends, lv, _ = synth.file_trace_table(n, seed)
        trace = ctx.synth_trace(n, seed, ends, lv, dtype=torch.int16)     # what read_abf's data section holds
        from pypore_amd import pipeline

        def step(k=None):
            st_, ln_, b, o, stt = pipeline.segment_file_trace(trace, synth.QUANTUM, params, threshold=90.0,
                                                              want_stats=args.stats)
            if k is not None and acc is not None:
                acc[k, :b.numel()].copy_(b)
                acc_counts[k] = b.numel()
            return b, o, stt
    torch.cuda.synchronize()

    # The final boundary-index gather (RCCL), as in Experiment.parse (results are collected after all files are
    # parsed): every rank writes the boundaries of its K batches into a [K, slot] send buffer and ONE all_gather (plus
    # one of the K counts) runs at the end of the job, inside the timed region.  (dist.BoundaryGather is the per-batch
    # form, one asynchronous collective per batch; it costs ~65 us of host time per batch.)
    acc, acc_counts, recv, recv_counts = None, None, None, None

    def final_gather():
        if not use_dist:
            return
        cnt = torch.from_numpy(acc_counts).to(acc.device)
        dist.all_gather_into_tensor(recv_counts, cnt)
        dist.all_gather_into_tensor(recv, acc.view(-1))

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if use_dist:                                         # slot size of the gather: twice the largest count seen
        b0, _, _ = step()
        most = max(int(t.numel()) for t in pdist.gather_varlen(b0))
        slot = 1 << int(np.ceil(np.log2(2 * most + 8)))
        acc = torch.zeros((args.steps, slot), dtype=torch.int32, device=b0.device)
        acc_counts = np.zeros(args.steps, dtype=np.int64)
        recv = torch.zeros(world * args.steps * slot, dtype=torch.int32, device=b0.device)
        recv_counts = torch.zeros(world * args.steps, dtype=torch.int64, device=b0.device)
    # A fresh process starts cold (GPU clocks, pinned staging buffers, the allocator's pools): settle for a fixed
    # 0.2 s before the W warmup steps so that a small W does not leak start-up effects into the K timed steps.
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < 0.2:
        step()
    for _ in range(args.warmup):
        step()
    kern = dict(blocksum_ms=0.0, spine_ms=0.0, bridge_ms=0.0, tree_ms=0.0, gather_ms=0.0, stitch_ms=0.0, seq_ms=0.0)
    seq_ms = 0.0
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        bounds, boff, _ = step(k)
        seq_ms += ctx.seq_ms()                           # HIP events on the library's stream: first upload .. last result copy
    final_gather()                                       # the job's boundary gather is inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    seq_ms /= args.steps
    # Per-kernel breakdown: a separate, untimed pass with an event between the phases (each such event keeps the next
    # kernel from starting back to back, ~6 us of idle GPU, so the timed region above runs without them).
    tm = ctx.timings()                                   # work counters of the last timed step
    ctx.set_option("timing", 2)
    for _ in range(0 if args.no_detail else args.steps):
        step()
        tm = ctx.timings()
        for k in kern:
            kern[k] += tm[k]
    ctx.set_option("timing", 1)
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        all_counts = recv_counts.view(world, args.steps).cpu().numpy()
        n_bounds = [int(c) for c in all_counts[:, -1]]            # the last batch's boundaries, all ranks
        mine = recv.view(world, args.steps, -1)[rank, args.steps - 1, :n_bounds[rank]]
        assert torch.equal(mine, bounds), "boundary gather returned something else for this rank"
    else:
        n_bounds = [int(bounds.numel())]
    for k in kern:
        kern[k] /= args.steps
    ms_per_step = dt / args.steps * 1e3
    value = world * n / (dt / args.steps) / 1e6

    if rank == 0:
        bytes_per_sample = BYTES_PER_SAMPLE if args.workload == "trace" else 2      # int16 counts in config 3
        # The path is a sequence of kernels; only K0 (blocksum) streams the samples, the scans work on its 2 B/sample
        # digest.  The roofline figure is therefore quoted for the whole sequence: algorithmic bytes of the path
        # (one read of every sample) over the duration of the sequence, measured with two HIP events on the
        # library's stream in the timed region (first upload .. last result copy).
        achieved = bytes_per_sample * n / (seq_ms * 1e-3) / 1e9 if seq_ms > 0 else 0.0
        names = ("blocksum_ms", "spine_ms", "bridge_ms", "tree_ms")
        dom = max(names, key=lambda k: kern[k])
        # per-kernel algorithmic bytes: K0 reads every sample and writes 16 B per 8-sample block; the scans read the
        # block prefix (16 B per block and window pass: two passes of overlapping windows on the spine) -- listed, not priced
        k0_bytes = (bytes_per_sample + 2) * n
        out = {
            "metric": "Msamples/sec segmented (SpeedyStatSplit, 10^8-sample trace); %HBM roofline",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("one %.0e-sample fp32 trace per GPU (5-level step signal, dwell U[1000,20000), "
                                    "sigma 1 pA, 2^-5 pA grid), single SpeedyStatSplit.parse over the whole trace; "
                                    "min_width=100 max_width=1e6 window_width=10000 prior_segments_per_second=10" % n)
                       if args.workload == "trace" else
                       ("BASELINE config 3: one %.0e-sample int16 .abf-shaped trace per GPU @100 kHz (110 pA open channel, "
                        "blockade events 1.5-10 s with dwells U[1000,20000)); lambda_event_parser(threshold=90) -> "
                        "per-event SpeedyStatSplit(prior_segments_per_second=10), end to end on the GPU" % n),
                       "samples_per_gpu": n, "boundaries": n_bounds, "segment_stats_in_step": bool(args.stats)},
            "roofline": {"bound": "hbm", "kernel": "kernel sequence of one ps_segment_batch (blocksum+spine+bridge+stitch+tree+gather)",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": round(achieved * 1e9 / HBM_PEAK, 5),
                         "traffic": PMC_TRAFFIC_BYTES if args.workload == "trace" else None,
                         "algorithmic_bytes_per_launch": bytes_per_sample * n,
                         "longest_kernel": dom.replace("_ms", "_kernel"),
                         "streaming_kernel": {"name": "blocksum_kernel", "ms": round(kern["blocksum_ms"], 4),
                                              "algorithmic_bytes": k0_bytes,
                                              "achieved": round(k0_bytes / (kern["blocksum_ms"] * 1e-3) / 1e9, 1) if kern["blocksum_ms"] > 0 else None,
                                              "frac": round(k0_bytes / (kern["blocksum_ms"] * 1e-3) / HBM_PEAK, 4) if kern["blocksum_ms"] > 0 else None,
                                              "traffic": PMC_TRAFFIC["blocksum_ms"] if args.workload == "trace" else None},
                         "traffic_per_kernel": {k.replace("_ms", ""): v for k, v in PMC_TRAFFIC.items()} if args.workload == "trace" else None,
                         "sequence_ms": round(seq_ms, 4),
                         "kernel_ms": {k: round(v, 4) for k, v in kern.items()},
                         "kernel_ms_note": "per-phase HIP events, separate untimed pass of the same steps (an event between "
                                           "two kernels costs ~6 us of idle GPU, so the timed region records only start and end)"},
            "whole_step_frac_of_hbm_roofline": round(bytes_per_sample * n / (ms_per_step * 1e-3) / HBM_PEAK, 5),
            "work": {k: tm[k] for k in ("windows", "candidates", "tiles", "tree_jobs", "repairs", "exact_rescans", "full_exact_scans")},
        }
        if not args.no_cpu:
            import oracle
            m = min(n, args.cpu_samples)
            x = trace[:m].cpu().numpy().astype(np.float64)
            if args.workload == "file":
                x = x * synth.QUANTUM
            t1 = time.perf_counter()
            if args.workload == "file":
                es, el = oracle.lambda_events(x, threshold=90.0)
                refs = [oracle.parse(x[a:a + l], **{k: v for k, v in PARAMS.items()}) for a, l in zip(es, el)]
                ref = np.concatenate(refs) if refs else np.zeros(0, np.int32)
            else:
                ref = oracle.parse(x, **{k: v for k, v in PARAMS.items()})
            t2 = time.perf_counter()
            got = bounds.cpu().numpy()
            # prefix property: parse(x[:m]) agrees with parse(x) away from the cut
            k = int(np.searchsorted(ref, m - 4 * PARAMS["window_width"])) if args.workload == "trace" else \
                int(sum(len(r) for r in refs[:-1]))
            out["cpu_baseline"] = {"value": round(m / (t2 - t1) / 1e6, 3), "unit": "Msamples/s", "cores": 1,
                                   "kind": "port",
                                   "sample": "first %d samples of rank 0's trace, oracle/statsplit_oracle.c "
                                             "(gcc -O2), single thread" % m,
                                   "prefix_boundaries_equal": bool(np.array_equal(ref[:k], got[:k]))}
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
