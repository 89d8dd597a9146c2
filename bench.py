#!/usr/bin/env python3
"""bench.py -- Msamples/s segmented by SpeedyStatSplit on a 10^8-sample trace (BASELINE.json).

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in HBM.
Workloads (`--workload`):

  trace (default)   BASELINE's metric: every rank segments its own 10^8-sample fp32 trace with one ps_segment_batch
                    (K0 block sums -> spine -> bridge -> stitch -> subtrees -> gather); weak scaling, no data-path
                    collective, one RCCL all_gather of the boundaries at the end of the K steps.
  file              BASELINE config 3: one 10^8-sample int16 .abf-shaped trace per rank, lambda_event_parser(threshold=90)
                    then per-event SpeedyStatSplit, end to end on the GPU.
  sharded-trace     BASELINE config 5: ONE 10^9-sample fp32 trace for the whole job; rank r holds [S_r, S_{r+1} + halo),
                    segments it as a stand-alone trace, the boundaries + spine flags are gathered (RCCL) and joined at
                    common spine anchors inside the timed region; strong scaling.  Rank 0 checks the result against the
                    same trace cut into 8 pieces (N = 1) or against its own whole piece (N > 1).
  files             BASELINE config 4: 64 int16 files of 7.5e7 samples, sharded over the ranks (longest first); every
                    file goes host (pinned) -> HBM on a copy stream while the previous file is detected + segmented on
                    the library's stream (two streams per GPU); boundaries gathered once at the end.  Inputs start in
                    HOST memory here, so this line is PCIe-inclusive by construction.

Prints ONE JSON line (rank 0).  `roofline` prices the kernel sequence of one step (two HIP events on the library's
stream) at the algorithmic bytes of the path against 8 TB/s; `roofline.traffic` is read from the PMC summary committed
under profiles/ (null when this run's arguments differ from the profiled ones); `cpu_baseline` times the CPU oracle
(oracle/, a port of the reference) on this host: one thread, and one thread per core over tiles of the trace.
"""
import argparse
import json
import os
import sys
import threading
import time

# HIP maps the streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The contexts in flight need one
# each: with RCCL initialised (N > 1, or one rank under torch.distributed) the communicator's streams take some, the
# contexts share what is left and the step of four contexts goes from 0.25 to 0.33 ms (tools/gpu_dist1.sh; with 8 queues
# 0.241 plain, 0.255 with RCCL).  The default since the end of round 4: sixteen contexts on 16 queues.
# Must be in the environment before the HIP runtime starts, i.e. before torch is imported.
# (a rank of an N > 1 job: RCCL's streams take about four queues -- sixteen contexts under an initialised communicator ran at
#  0.197 / 0.184 / 0.184 ms per step on 16 / 20 / 24 queues, tools/gpu_r4_rccl_queues.sh; without RCCL 16 is the steadier choice)
_QUEUES_FROM_CALLER = "GPU_MAX_HW_QUEUES" in os.environ
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20" if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1 else "16")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PARAMS = dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10.,
              sampling_freq=1e5)
HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md
PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")     # written by tools/profile_round.sh from the PMC passes
METRIC = "Msamples/sec segmented (SpeedyStatSplit, 10^8-sample trace); %HBM roofline"


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_parse_tiled(x, cores, halo):
    """The oracle on `cores` threads: contiguous pieces with a halo, each parsed as a stand-alone trace (ctypes releases
    the GIL), joined at common spine anchors -- the same decomposition the multi-GPU path uses."""
    import oracle
    from pypore_amd import dist as pdist
    n = x.size
    ranges = pdist.shard_ranges(n, cores, halo)
    res = [None] * cores

    def work(r):
        lo, hi = ranges[r]
        res[r] = oracle.parse_flags(x[lo:hi], **PARAMS)

    th = [threading.Thread(target=work, args=(r,)) for r in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    pieces = [(ranges[r][0], ranges[r][1], res[r][0], res[r][1]) for r in range(cores)]
    return pdist.stitch_pieces(pieces, n, PARAMS["window_width"], PARAMS["min_width"])


# ---- the N > 1 pieces of the bench, factored out of main() so that tests/test_bench_contract.py can drive them over gloo
#      on CPU tensors (torch.distributed is backend-agnostic: on the GPU box the same code runs on nccl = RCCL) ------------
class JobGather:
    """The one boundary gather of a trace / file job (as in Experiment.parse, which collects after all files are parsed):
    every rank writes the boundaries of its K batches into row k of a [K, slot] send buffer and ONE all_gather of the
    buffer plus one of the K counts runs at the end of the job, inside the timed region."""

    def __init__(self, steps, slot, world, device):
        import torch
        self.steps, self.slot, self.world, self.device = steps, slot, world, device
        self.acc = torch.zeros((steps, slot), dtype=torch.int32, device=device)
        self.counts = np.zeros(steps, dtype=np.int64)
        self.recv = torch.zeros(world * steps * slot, dtype=torch.int32, device=device)
        self.recv_counts = torch.zeros(world * steps, dtype=torch.int64, device=device)

    @staticmethod
    def slot_for(most):
        return 1 << int(np.ceil(np.log2(2 * most + 8)))

    def put(self, k, b):
        """batch k's boundaries (for callers that do not write into acc[k] themselves)"""
        self.acc[k, :b.numel()].copy_(b)
        self.counts[k] = b.numel()

    def run(self):
        import torch
        import torch.distributed as dist
        cnt = torch.from_numpy(self.counts).to(self.device)
        dist.all_gather_into_tensor(self.recv_counts, cnt)
        dist.all_gather_into_tensor(self.recv, self.acc.view(-1))

    def last_counts(self):
        """boundaries of the job's last batch, per rank"""
        return [int(c) for c in self.recv_counts.view(self.world, self.steps).cpu().numpy()[:, -1]]

    def row(self, rank, k):
        cnt = int(self.recv_counts.view(self.world, self.steps)[rank, k])
        return self.recv.view(self.world, self.steps, -1)[rank, k, :cnt]


def sharded_trace_join(b, sp, ranges, n, W, mw, halo, repair):
    """config 5, one step: this rank's piece boundaries + spine flags -> everybody's (two variable-length gathers) ->
    joined at common spine anchors (every rank walks the same gathered data)"""
    from pypore_amd import dist as pdist
    allb = pdist.gather_varlen(b)
    allf = pdist.gather_varlen(sp)
    pieces = [(ranges[r][0], ranges[r][1], allb[r].cpu().numpy(), allf[r].cpu().numpy()) for r in range(len(ranges))]
    return pdist.stitch_pieces(pieces, n, W, mw, repair=repair, halo=halo)


def sharded_trace_repair(rank, r_up, piece_fn, device):
    """a seam without a common anchor: the upstream rank re-segments the stretch, everybody receives it"""
    import torch
    from pypore_amd import dist as pdist
    if rank == r_up:
        rb, rf = piece_fn()
    else:
        rb = torch.zeros(0, dtype=torch.int32, device=device)
        rf = torch.zeros(0, dtype=torch.uint8, device=device)
    gb, gf = pdist.gather_varlen(rb), pdist.gather_varlen(rf)
    return gb[r_up].cpu().numpy(), gf[r_up].cpu().numpy()


def files_job_gather(bounds_per_file, device):
    """config 4: the job's one gather -- the per-file counts, then the concatenated boundaries; returns per rank
    (counts, boundaries)"""
    import torch
    from pypore_amd import dist as pdist
    cat = torch.cat(bounds_per_file) if bounds_per_file else torch.zeros(0, dtype=torch.int32, device=device)
    cnts = pdist.gather_varlen(torch.tensor([b.numel() for b in bounds_per_file], dtype=torch.int32, device=device))
    return cnts, pdist.gather_varlen(cat)


def max_over_ranks(dt, device):
    """(MAX over ranks of the job's time, every rank's time) -- the contract's clock"""
    import torch
    import torch.distributed as dist
    mine = torch.tensor([dt], dtype=torch.float64, device=device)
    every = torch.zeros(dist.get_world_size(), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(every, mine)
    return float(every.max().item()), [float(x) for x in every.cpu().numpy()]


def self_launch(argv, n):
    """`python bench.py --gpus N` (N > 1) started as ONE process: start the N ranks as children under
    torch.distributed.run, BEFORE anything here touches the GPU (no exec: a process that initialised HIP must not be
    replaced), pass rank 0's JSON line through (the children inherit stdout) and leave with their exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not _QUEUES_FROM_CALLER:
        env.pop("GPU_MAX_HW_QUEUES", None)               # the ranks choose for themselves (N > 1: room for RCCL's streams)
    return subprocess.run(cmd, env=env).returncode


def selftest_dist(args):
    """--selftest-dist: the N > 1 plumbing of this file on CPU tensors over gloo (no GPU, no library): the job gather,
    the sharded-trace join with a synthetic per-rank segmenter, the files gather, the clock.  Rank 0 prints one JSON line."""
    import torch
    import torch.distributed as dist
    from pypore_amd import dist as pdist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == args.gpus, (world, args.gpus)
    dist.init_process_group("gloo")
    dev = torch.device("cpu")
    ok = {}
    # job gather: K batches of rank-dependent boundaries
    K = 5
    mk = lambda r, k: torch.arange(3 + 2 * r + k, dtype=torch.int32) * (r + 1) + k                 # noqa: E731
    jg = JobGather(K, JobGather.slot_for(3 + 2 * world + K), world, dev)
    for k in range(K):
        jg.put(k, mk(rank, k))
    jg.run()
    ok["job_gather"] = all(torch.equal(jg.row(r, k), mk(r, k)) for r in range(world) for k in range(K)) \
        and jg.last_counts() == [3 + 2 * r + K - 1 for r in range(world)]
    # sharded trace: boundaries every 700 samples, every one a spine anchor; a piece reports those inside it
    n, W, mw = 200000 if world <= 4 else 480000, 2000, 100
    halo = 8 * W
    ranges = pdist.shard_ranges(n, world, halo)
    truth = np.arange(700, n, 700, dtype=np.int64)

    def piece(lo, hi):
        b = truth[(truth >= lo + 1) & (truth < hi - mw)] - lo
        return torch.from_numpy(b.astype(np.int32)), torch.ones(b.size, dtype=torch.uint8)

    lo, hi = ranges[rank]
    b, sp = piece(lo, hi)
    got = sharded_trace_join(b, sp, ranges, n, W, mw, halo,
                             lambda r_up, lo2, hi2: sharded_trace_repair(rank, r_up, lambda: piece(lo2, hi2), dev))
    ok["sharded_trace_join"] = bool(np.array_equal(np.asarray(got, dtype=np.int64), truth[truth < n - mw]))
    # the same with a FORCED seam repair (round 5): no boundary from 3 000 samples before the second piece's start to 3 000
    # behind the end of its upstream neighbour's halo -- the two pieces share no anchor, the upstream rank re-segments a
    # longer stretch until it meets a downstream chain (dist.stitch_pieces: extend and re-run that seam only)
    if world > 1:
        s1 = ranges[1][0]
        truth2 = truth[(truth < s1 - 3000) | (truth > s1 + halo + 3000)]
        repairs = [0]

        def piece2(lo_, hi_):
            b_ = truth2[(truth2 >= lo_ + 1) & (truth2 < hi_ - mw)] - lo_
            return torch.from_numpy(b_.astype(np.int32)), torch.ones(b_.size, dtype=torch.uint8)

        def repair2(r_up, lo2, hi2):
            repairs[0] += 1
            return sharded_trace_repair(rank, r_up, lambda: piece2(lo2, hi2), dev)

        b2, sp2 = piece2(lo, hi)
        got2 = sharded_trace_join(b2, sp2, ranges, n, W, mw, halo, repair2)
        ok["sharded_trace_join_with_seam_repair"] = bool(np.array_equal(np.asarray(got2, dtype=np.int64), truth2[truth2 < n - mw])) \
            and repairs[0] >= 1
    # files gather; BASELINE config 4's shard: 64 files over the ranks (shard_units, longest first), every file exactly once
    n_units = 64 if world >= 8 else 7
    unit_len = [1000 + 10 * ((7 * f) % n_units) for f in range(n_units)]
    shards = pdist.shard_units(unit_len, world)
    mine = shards[rank]
    cnts, cat = files_job_gather([torch.full((f % 9 + 1,), f, dtype=torch.int32) for f in mine], dev)
    ok["files_gather"] = all([int(c) for c in cnts[r]] == [f % 9 + 1 for f in shards[r]] and
                             [int(x) for x in cat[r]] == [f for f in shards[r] for _ in range(f % 9 + 1)] for r in range(world))
    loads = [sum(unit_len[f] for f in sh) for sh in shards]
    ok["files_shard_covers_every_unit_once"] = sorted(f for sh in shards for f in sh) == list(range(n_units))
    ok["files_shard_balanced"] = max(loads) <= 1.1 * (sum(loads) / world) + max(unit_len)
    # dist.BoundaryGather with the C ABI's gather (backend "library", ps_gather_bounds: every rank sends `capacity` int32,
    # slot r of the result at r * capacity) against the torch backend, batch by batch: the header, the in-place send of a
    # buffer that already holds header + payload, a contribution that does not fit on ONE rank (every rank sees it in the
    # gathered counts and takes the two-collective fall-back together), an empty contribution.  The transport here is a
    # stand-in with ps_gather_bounds' contract over gloo (RCCL needs GPUs); the slot layout is the code that runs on them.
    class GlooSlots(object):
        world = dist.get_world_size()

        def gather_bounds(self, send, recv, stream=None):
            assert send.dtype == torch.int32 and recv.numel() == self.world * send.numel()
            dist.all_gather_into_tensor(recv, send.contiguous())

    cap = 64
    bg_lib = pdist.BoundaryGather(cap, dev, backend="library", comm=GlooSlots())
    bg_tor = pdist.BoundaryGather(cap, dev, backend="torch")
    H = pdist.BoundaryGather.HEADER

    def contribution(r, batch):
        if batch == 0:
            return torch.arange(5 + r, dtype=torch.int32) * 7 + r                       # fits
        if batch == 1:
            return torch.arange((cap + 9) if r == world - 1 else 3, dtype=torch.int32) - r    # the last rank overflows
        if batch == 2:
            return torch.zeros(0 if r % 2 else 4, dtype=torch.int32) + r                  # empty on odd ranks
        buf = torch.zeros(cap + 16, dtype=torch.int32)                                  # header + payload already laid out
        view = buf[H:H + 6 + r]
        view.copy_(torch.arange(6 + r, dtype=torch.int32) * 3 - r)
        return view
    same, layout = True, True
    for batch in range(4):
        mine_b = contribution(rank, batch)
        t_lib = bg_lib.submit(mine_b)
        got_lib = [x.clone() for x in bg_lib.result(t_lib)]
        t_tor = bg_tor.submit(mine_b.clone() if batch < 3 else contribution(rank, batch))
        got_tor = [x.clone() for x in bg_tor.result(t_tor)]
        want = [contribution(r, batch) for r in range(world)]
        same = same and all(torch.equal(a, b) for a, b in zip(got_lib, got_tor))
        layout = layout and all(torch.equal(a, b) for a, b in zip(got_lib, want))
    # the raw rows (host=False): count in column 0, payload from HEADER on, an overflow visible as a count beyond the slot
    t_lib = bg_lib.submit(contribution(rank, 1))
    rows = bg_lib.result(t_lib, host=False)
    counts_seen = [int(c) for c in rows[:, 0].tolist()]
    ok["boundary_gather_library_equals_torch_backend"] = bool(same)
    ok["boundary_gather_library_slot_layout"] = bool(layout) and counts_seen == [(cap + 9) if r == world - 1 else 3 for r in range(world)] \
        and rows.shape == (world, cap) and max(counts_seen) > cap - H
    bg_lib.close()
    bg_tor.close()
    tmax, every = max_over_ranks(0.001 * (rank + 1), dev)
    ok["clock"] = abs(tmax - 0.001 * world) < 1e-12 and len(every) == world
    if rank == 0:
        print(json.dumps({"selftest": "dist", "ok": bool(all(ok.values())), "checks": ok, "ranks_seen": dist.get_world_size(),
                          "backend": dist.get_backend(), "files_units": n_units,
                          "shard_imbalance": round(max(loads) / (sum(loads) / world), 4)}))
    dist.destroy_process_group()
    return 0 if all(ok.values()) else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--samples", type=int, default=None, help="samples per trace / file (default: the workload's BASELINE size)")
    ap.add_argument("--files", type=int, default=64, help="files of the job (workload files)")
    ap.add_argument("--cpu-samples", type=int, default=100_000_000, help="samples of rank 0's trace the CPU baseline runs on")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the PCIe-inclusive measurement (trace workload)")
    ap.add_argument("--no-detail", action="store_true", help="skip the separate per-kernel timing pass (kernel_ms stays 0)")
    ap.add_argument("--stats", action="store_true", help="also run the per-segment statistics kernel in the step")
    ap.add_argument("--dwell", type=int, nargs=2, default=None, metavar=("LO", "HI"),
                    help="dwell range of the synthetic trace in samples (default: BASELINE's U[1000, 20000))")
    ap.add_argument("--streams", type=int, default=16,
                    help="contexts (HIP streams) the K steps of the trace / file workloads are spread over: independent "
                         "batches overlap on the GPU (engine.StreamPool); 1 = one batch at a time")
    ap.add_argument("--workload", choices=["trace", "file", "sharded-trace", "files"], default="trace")
    ap.add_argument("--diag-env", action="store_true",
                    help="experiments only: apply the PORESEG_* variables of the environment as context options "
                         "(engine.apply_env_defaults); without it they are recorded in host.poreseg_env_seen and ignored")
    ap.add_argument("--selftest-dist", action="store_true",
                    help="run only the N > 1 plumbing of this file on CPU tensors over gloo (no GPU needed) and print one JSON line")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by itself: become the launcher of the N ranks (before torch or the GPU are touched)
        sys.exit(self_launch(sys.argv[1:], args.gpus))
    if args.selftest_dist:
        if "WORLD_SIZE" not in os.environ:               # --gpus 1: a world of one
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
        sys.exit(selftest_dist(args))
    wl = args.workload
    defaults = {"trace": (100, 20, 100_000_000), "file": (100, 20, 100_000_000),
                "sharded-trace": (20, 5, 1_000_000_000), "files": (3, 1, 75_000_000)}[wl]
    steps = defaults[0] if args.steps is None else args.steps
    warmup = defaults[1] if args.warmup is None else args.warmup
    n = defaults[2] if args.samples is None else args.samples

    import torch
    import torch.distributed as dist
    from pypore_amd import _lib, engine, synth
    from pypore_amd import dist as pdist

    rank = int(os.environ.get("RANK", "0"))

    def stage(msg):
        """progress of an N > 1 job on stderr (stdout carries rank 0's one JSON line)"""
        sys.stderr.write("[bench] %s\n" % msg)
        sys.stderr.flush()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = args.gpus > 1 or world > 1 or os.environ.get("PORESEG_BENCH_DIST") == "1"   # env: exercise the
    if use_dist:                                         # N > 1 code path on one GPU (torchrun --nproc-per-node 1)
        assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        stage("rank %d/%d: init_process_group(nccl) on cuda:%d of %d visible" % (rank, world, local_rank, torch.cuda.device_count()))
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        # a first run on N GPUs that dies should say where: every rank reports on stderr before the clock
        try:
            rccl = ".".join(str(v_) for v_ in torch.cuda.nccl.version())
        except Exception as e_:                          # noqa: BLE001
            rccl = "unknown (%s)" % e_
        probe = torch.tensor([rank], dtype=torch.int64, device=torch.device("cuda", local_rank))
        seen = torch.zeros(world, dtype=torch.int64, device=probe.device)
        dist.all_gather_into_tensor(seen, probe)
        stage("rank %d: RCCL %s, ranks_seen=%d, first all_gather %s" % (rank, rccl, dist.get_world_size(), seen.cpu().tolist()))
    else:
        torch.cuda.set_device(0)
    dev = torch.cuda.current_device()
    device = torch.device("cuda", dev)
    if args.diag_env:
        engine.apply_env_defaults()
    ctx = engine.context(dev)
    params = _lib.split_params(**PARAMS)
    W, mw = PARAMS["window_width"], PARAMS["min_width"]

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    T = max(1, args.streams) if wl in ("trace", "file") else 1
    pool = engine.StreamPool(dev, T)
    acc = acc_counts = None
    jg = None                                            # the job's boundary gather (N > 1)
    check = {}                                           # parity evidence gathered outside the timed region
    final_gather = lambda: None                          # noqa: E731
    trace = None
    seed = 2024 + (rank if wl in ("trace", "file") else 0)      # rank 0's trace is golden case G7

    # ------------------------------------------------------------------------------------------------ workloads
    if wl in ("trace", "file"):
        if wl == "trace":
            # Independent batches are independent DATA: stream t segments its own trace (generator seed + 1000 t; stream 0
            # of rank 0 is golden case G7, seed 2024).  T x 0.4 GB of samples resident in HBM.
            traces = []
            for t_ in range(T):
                sd = seed + 1000 * t_
                d = synth.dwell_table(sd, n, *args.dwell) if args.dwell else synth.dwell_table(sd, n)
                ends = np.cumsum(d)
                lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
                traces.append(ctx.synth_trace(n, sd, ends, lv, dtype=torch.float32))
            trace = traces[0]
            ev_off = np.array([0, n], dtype=np.int64)
            outs = [torch.empty(n // mw + 1, dtype=torch.int32, device=device) for _ in range(T)]   # reused result buffers

            def step(k=None, cx=None, t=0):
                # k: index of a timed batch when the job gathers its boundaries at the end (N > 1): they are written
                # straight into row k of the send buffer
                b, o, st = (cx or ctx).segment_batch(traces[t], ev_off, params, synth.QUANTUM, want_stats=args.stats,
                                                     out=acc[k] if k is not None and acc is not None else outs[t])
                if k is not None and acc is not None:
                    acc_counts[k] = b.numel()
                return b
        else:
            ends, lv, _ = synth.file_trace_table(n, seed)
            trace = ctx.synth_trace(n, seed, ends, lv, dtype=torch.int16)     # what read_abf's data section holds
            from pypore_amd import pipeline

            def step(k=None, cx=None, t=0):
                st_, ln_, b, o, stt = pipeline.segment_file_trace(trace, synth.QUANTUM, params, threshold=90.0,
                                                                  want_stats=args.stats, ctx=cx)
                if k is not None and acc is not None:
                    acc[k, :b.numel()].copy_(b)
                    acc_counts[k] = b.numel()
                return b
        samples_per_step = n * world
        bytes_per_sample = 4 if wl == "trace" else 2
        scaling = "weak"
        if use_dist:
            # The final boundary-index gather (RCCL), as in Experiment.parse (results are collected after all files are
            # parsed): every rank writes the boundaries of its K batches into a [K, slot] send buffer and ONE all_gather
            # (plus one of the K counts) runs at the end of the job, inside the timed region.
            b0 = step()
            most = max(int(t.numel()) for t in pdist.gather_varlen(b0))
            jg = JobGather(steps, JobGather.slot_for(most), world, device)
            acc, acc_counts = jg.acc, jg.counts
            final_gather = jg.run
            stage("rank %d: first batch %d boundaries; per_rank_boundaries before the clock %s; gather slot %d"
                  % (rank, int(b0.numel()), [int(t.numel()) for t in pdist.gather_varlen(b0)], jg.slot))

    elif wl == "sharded-trace":
        halo = 8 * W
        d = synth.dwell_table(seed, n, *args.dwell) if args.dwell else synth.dwell_table(seed, n)
        ends = np.cumsum(d)
        lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
        ranges = pdist.shard_ranges(n, world, halo)
        lo, hi = ranges[rank]
        trace = ctx.synth_trace(hi - lo, seed, ends, lv, dtype=torch.float32, start=lo)
        piece_off = np.array([0, hi - lo], dtype=np.int64)
        out1 = torch.empty((hi - lo) // mw + 1, dtype=torch.int32, device=device)

        def segment_piece(t, off, out=None):
            b, o, st, sp = ctx.segment_batch(t, off, params, synth.QUANTUM, want_stats=False, want_spine=True, out=out)
            return b, sp

        def regen_segment(lo2, hi2):
            # a seam without a common anchor inside the halo (long dwells): the stretch is generated and segmented afresh
            t2 = ctx.synth_trace(hi2 - lo2, seed, ends, lv, dtype=torch.float32, start=lo2)
            return segment_piece(t2, np.array([0, hi2 - lo2], dtype=np.int64))

        if use_dist:
            def repair(r_up, lo2, hi2):
                return sharded_trace_repair(rank, r_up, lambda: regen_segment(lo2, hi2), device)

            def step(k=None):
                b, sp = segment_piece(trace, piece_off, out1)
                return sharded_trace_join(b, sp, ranges, n, W, mw, halo, repair)
        else:
            def step(k=None):
                b, sp = segment_piece(trace, piece_off, out1)
                return b
        samples_per_step = n
        bytes_per_sample = 4
        scaling = "strong"

    else:                                                # files (config 4)
        n_files = args.files
        lens = [n] * n_files
        shards = pdist.shard_units(lens, world)
        mine = shards[rank]
        # file contents: four distinct synthetic int16 traces in pinned host memory, file f uses table f % 4 (the
        # boundaries differ per table; what matters for the bench is the data path: host -> HBM -> kernels)
        n_tables = min(4, max(1, len(mine)))
        host = []
        for t in range(n_tables):
            ends_t, lv_t, _ = synth.file_trace_table(n, 7000 + t)
            dtrace = ctx.synth_trace(n, 7000 + t, ends_t, lv_t, dtype=torch.int16)
            h = torch.empty(n, dtype=torch.int16, pin_memory=True)
            h.copy_(dtrace)
            host.append(h)
            del dtrace
        torch.cuda.synchronize()
        dbuf = [torch.empty(n, dtype=torch.int16, device=device) for _ in range(2)]
        copy_stream = torch.cuda.Stream(device=device)
        from pypore_amd import pipeline

        def step(k=None):
            results = []
            evs = [torch.cuda.Event() for _ in mine]
            if mine:
                with torch.cuda.stream(copy_stream):
                    dbuf[0].copy_(host[mine[0] % n_tables], non_blocking=True)
                    evs[0].record(copy_stream)
            for j, f in enumerate(mine):
                evs[j].synchronize()                     # file j is in HBM
                if j + 1 < len(mine):                    # file j+1 travels while file j is segmented (the buffer it
                    with torch.cuda.stream(copy_stream):  # overwrites was consumed by the blocking call of file j-1)
                        dbuf[(j + 1) % 2].copy_(host[mine[j + 1] % n_tables], non_blocking=True)
                        evs[j + 1].record(copy_stream)
                st_, ln_, b, o, _ = pipeline.segment_file_trace(dbuf[j % 2], synth.QUANTUM, params, threshold=90.0)
                results.append((st_, ln_, b.clone(), o))
            if use_dist:                                 # the job's one gather: counts, then the padded payload
                files_job_gather([r[2] for r in results], device)
            return results
        samples_per_step = n * n_files
        bytes_per_sample = 2
        scaling = "strong"
        # how evenly shard_units dealt the files (longest-first greedy, DataTypes.py:968-984 is the shard axis)
        loads = [sum(lens[f] for f in sh) for sh in shards]
        check["files_per_rank"] = [len(sh) for sh in shards]
        check["shard_imbalance"] = round(max(loads) / (sum(loads) / len(loads)), 4) if sum(loads) else 1.0

    torch.cuda.synchronize()
    # CPython's cyclic collector is stop-the-world: a full collection walks every object torch and numpy created at
    # import (40 ms here) while holding the GIL, and every host thread of the pool then waits for it on its way out of
    # the C call (tools/pool_stalls.py: all contexts stall together, the device sequences stay at 1.2 ms).  A
    # long-running host does what is done here: collect once, then move what is alive out of the collector's reach.
    # (Before the warm-up, not after it: the GPU drops its clocks during a pause of that length.)
    import gc
    gc.collect()
    gc.freeze()
    # A fresh process starts cold (GPU clocks, pinned staging buffers, the allocator's pools): settle for a fixed
    # 0.2 s before the W warmup steps so that a small W does not leak start-up effects into the K timed steps.
    # (With several contexts the settling runs on the pool, like the timed steps: every context sizes its scratch, every host
    #  thread is awake and the GPU is under the load of the timed region when the clock starts.  Round 4: with the settling
    #  and the W steps on one context only, one in six fresh processes with twelve contexts ran its 20 timed steps at twice
    #  the usual time; 60 consecutive runs inside one process never did: tools/pool_short_runs.py.)
    warm = (lambda m: pool.run(m, lambda cx, k, t: step(None, cx, t))) if T > 1 else (lambda m: [step() for _ in range(m)])
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < 0.2:
        warm(max(T, 1))
    if warmup:
        warm(warmup)                                     # the W untimed steps of the contract: the same steps as the timed ones
    kern = dict(blocksum_ms=0.0, spine_ms=0.0, bridge_ms=0.0, tree_ms=0.0, gather_ms=0.0, stitch_ms=0.0, seq_ms=0.0)
    seq_ms = 0.0
    if T > 1:
        warm(2 * T)                                      # every context has sized its scratch before the clock starts
    seq_acc = [0.0] * T

    def timed(cx, k, t):
        r = step(k, cx, t) if T > 1 else step(k)
        seq_acc[t] += cx.seq_ms()                        # HIP events on the context's stream: first upload .. last result copy
        return r

    # (the job's gather once before the clock starts, like the W warm-up steps of the hot path: the first collective of a
    #  shape sets up RCCL's channels and buffers -- 9 ms, which a 100-step run showed as +0.09 ms per step under
    #  torch.distributed and a 20-step run as +0.4)
    final_gather()
    barrier()
    t0 = time.perf_counter()
    results = pool.run(steps, timed)                     # K steps, step k on stream k % T (T = 1: one after the other)
    final_gather()                                       # the job's boundary gather is inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    gc.unfreeze()
    result = results[-1]
    seq_ms = sum(seq_acc) / steps
    # one batch at a time on one stream (the latency of a single call), outside the timed region
    single = None
    if T > 1:
        torch.cuda.synchronize()
        k1 = min(steps, 30)
        s1 = 0.0
        t1 = time.perf_counter()
        for _ in range(k1):
            step()
            s1 += ctx.seq_ms()
        torch.cuda.synchronize()
        single = ((time.perf_counter() - t1) / k1 * 1e3, s1 / k1)
    tm = ctx.timings()                                   # work counters of the last step on the first context
    # Per-kernel breakdown: a separate, untimed pass with an event between the phases (each such event keeps the next
    # kernel from starting back to back, ~6 us of idle GPU, so the timed region above runs without them).
    detail = wl in ("trace", "file", "sharded-trace") and not args.no_detail
    ctx.set_option("timing", 2)
    for _ in range(min(steps, 20) if detail else 0):
        step()
        t2 = ctx.timings()
        for k in kern:
            kern[k] += t2[k]
    ctx.set_option("timing", 1)
    for k in kern:
        kern[k] /= max(1, min(steps, 20))
    per_rank_ms = [dt / steps * 1e3]
    if use_dist:
        dt, every = max_over_ranks(dt, device)
        per_rank_ms = [t_ / steps * 1e3 for t_ in every]
    ms_per_step = dt / steps * 1e3
    value = samples_per_step / (dt / steps) / 1e6

    # ------------------------------------------------------------------------------------------------ parity evidence
    if wl in ("trace", "file"):
        bounds = result
        if wl == "trace" and rank == 0:
            import hashlib
            # (a) the results the T streams produced in flight == the same traces segmented one call at a time on context 0
            last = {}
            for k_, r_ in enumerate(results):
                last[k_ % T] = r_.clone() if T > 1 else r_
            same = True
            per_stream = []
            for t_ in sorted(last):
                one, _, _ = ctx.segment_batch(traces[t_], ev_off, params, synth.QUANTUM, want_stats=False)
                same = same and bool(torch.equal(one, last[t_]))
                per_stream.append(int(one.numel()))
            check["all_streams_equal_single_stream"] = same
            check["boundaries_per_stream"] = per_stream
            bounds = last[0]
            # (b) stream 0's trace is golden case G7 (tests/golden/manifest.json: SHA-256 of the reference's boundaries)
            if n == 100_000_000 and not args.dwell and seed == 2024:
                try:
                    with open(os.path.join(ROOT, "tests", "golden", "manifest.json")) as f:
                        g7 = [c for c in json.load(f)["cases"] if c["name"] == "G7_1e8"][0]
                    b0 = last[0].cpu().numpy().astype(np.int32)
                    check["g7_sha256_equal"] = bool(hashlib.sha256(b0.tobytes()).hexdigest() == g7["sha256"] and len(b0) == g7["n_bounds"])
                except (OSError, IndexError, KeyError):
                    check["g7_sha256_equal"] = None
            assert check["all_streams_equal_single_stream"] and check.get("g7_sha256_equal", True) is not False, check
        if use_dist:
            n_bounds = jg.last_counts()                               # the last batch's boundaries, all ranks
            # (the last batch ran on stream (steps - 1) % T, i.e. on that stream's trace)
            assert torch.equal(jg.row(rank, steps - 1), results[-1][:n_bounds[rank]]), "boundary gather returned something else for this rank"
        else:
            n_bounds = [int(bounds.numel())]
    elif wl == "sharded-trace":
        got = np.asarray(result.cpu().numpy() if hasattr(result, "cpu") else result, dtype=np.int64)
        n_bounds = [int(got.size)]
        if rank == 0:
            if world == 1:
                # the same trace as 8 stand-alone pieces with halo on this GPU, joined at common spine anchors
                pr = pdist.shard_ranges(n, 8, halo)
                pieces = []
                for (plo, phi) in pr:
                    b, sp = segment_piece(trace[plo:phi], np.array([0, phi - plo], dtype=np.int64))
                    pieces.append((plo, phi, b.cpu().numpy(), sp.cpu().numpy()))
                ref = pdist.stitch_pieces(pieces, n, W, mw, halo=halo,
                                          repair=lambda r_, lo2, hi2: tuple(x_.cpu().numpy() for x_ in regen_segment(lo2, hi2)))
                check["whole_trace_equals_8_stitched_pieces"] = bool(np.array_equal(ref, got))
                check["seam_repairs"] = int(pdist.stitch_pieces.last_repairs)
            else:
                b, sp = segment_piece(trace, piece_off)
                mine_g = b.cpu().numpy().astype(np.int64) + lo
                lim = ranges[0][1] - 2 * W - 2 * mw
                check["rank0_piece_prefix_equal"] = bool(np.array_equal(mine_g[mine_g <= lim], got[got <= lim]))
            assert all(v for v in check.values() if isinstance(v, bool)), check
    else:
        n_bounds = [int(sum(r[2].numel() for r in result))]
        if rank == 0 and mine:
            # every distinct table the step segmented == the same file segmented on its own (fresh upload, one call, outside
            # the timed region); the first events of the first table also against the CPU restatement of the reference
            from pypore_amd import pipeline as _pl
            ok_tables = {}
            for j, f in enumerate(mine):
                tb = f % n_tables
                if tb in ok_tables:
                    continue
                one = torch.empty(n, dtype=torch.int16, device=device)
                one.copy_(host[tb])
                st1, ln1, b1, o1, _ = _pl.segment_file_trace(one, synth.QUANTUM, params, threshold=90.0)
                st0, ln0, b0, o0 = result[j]
                ok_tables[tb] = bool(np.array_equal(st0, st1) and np.array_equal(ln0, ln1) and torch.equal(b0, b1) and np.array_equal(o0, o1))
                if tb == mine[0] % n_tables and not args.no_cpu:
                    import oracle
                    x0 = host[tb].numpy()
                    okc, took = True, 0
                    for e in range(min(3, len(st1))):
                        a_, l_ = int(st1[e]), int(ln1[e])
                        ref = oracle.parse(x0[a_:a_ + l_].astype(np.float64) * synth.QUANTUM, **PARAMS)
                        okc = okc and bool(np.array_equal(ref, b1[int(o1[e]):int(o1[e + 1])].cpu().numpy()))
                        took += 1
                    check["first_events_equal_cpu_oracle"] = okc
                    check["events_checked_against_cpu_oracle"] = took
                del one
            check["tables_equal_standalone_segmentation"] = ok_tables
            check["events_per_table"] = {int(f % n_tables): int(len(result[j][0])) for j, f in enumerate(mine)}
            assert all(ok_tables.values()) and check.get("first_events_equal_cpu_oracle", True), check

    if rank == 0:
        # (files: many calls per step, PCIe-inclusive; several streams: calls overlap, the job's clock is the measure)
        seq = seq_ms if (wl != "files" and T == 1) else ms_per_step
        per_gpu_bytes = bytes_per_sample * samples_per_step / world
        achieved = per_gpu_bytes / (seq * 1e-3) / 1e9 if seq > 0 else 0.0
        names = ("blocksum_ms", "spine_ms", "bridge_ms", "tree_ms")
        dom = max(names, key=lambda k: kern[k])
        traffic = None
        if wl == "trace" and os.path.exists(PMC_FILE) and not args.dwell and not args.stats and n == 100_000_000 \
                and not os.environ.get("PORESEG_LIB") and not args.diag_env:
            with open(PMC_FILE) as f:
                traffic = json.load(f)                   # {"source": ..., "total": bytes, "per_kernel": {...}}
        # what K0 moves per launch: the samples once + 8 B of digest per 8 samples + 16 B of group record per 256 + 16 B of
        # chunk totals per 1 024 = bytes_per_sample + 1.078 B per sample -- or, where this run is the profiled
        # command, the bytes the PMC passes counted for the kernel (round 5's line priced it at +2 B: round 2's 16-byte digest)
        n_k0 = trace.numel() if trace is not None else n
        k0_bytes = int(round((bytes_per_sample + 1.0 + 16.0 / 256.0 + 16.0 / 1024.0) * n_k0))
        k0_bytes_src = "model: %d B read + 1.078 B written per sample (digest 8 B / 8 samples, group record 16 B / 256, chunk totals 16 B / 1 024)" % bytes_per_sample
        if traffic and traffic.get("per_kernel", {}).get("blocksum_kernel"):
            k0_bytes = int(traffic["per_kernel"]["blocksum_kernel"])
            k0_bytes_src = "PMC: FETCH x 2 + WRITE of the kernel, " + os.path.basename(PMC_FILE)
        workload_text = {
            "trace": "one %.0e-sample fp32 trace per GPU (5-level step signal, dwell U[1000,20000), sigma 1 pA, 2^-5 pA grid), "
                     "single SpeedyStatSplit.parse over the whole trace; every stream segments its own trace (seeds 2024 + 1000 t)" % n,
            "file": "BASELINE config 3: one %.0e-sample int16 .abf-shaped trace per GPU @100 kHz (110 pA open channel, blockade "
                    "events 1.5-10 s with dwells U[1000,20000)); lambda_event_parser(threshold=90) -> per-event "
                    "SpeedyStatSplit, end to end on the GPU" % n,
            "sharded-trace": "BASELINE config 5: ONE %.0e-sample fp32 trace for the job, rank r segments [S_r, S_r+1 + 8 W) as a "
                             "stand-alone trace, boundaries + spine flags gathered and joined at common spine anchors inside "
                             "the timed region" % n,
            "files": "BASELINE config 4: %d int16 files of %.1e samples (.abf data sections in pinned host memory) sharded over "
                     "the ranks; per file host->HBM on a copy stream overlapped with lambda_event_parser + SpeedyStatSplit of "
                     "the previous file; PCIe-inclusive" % (args.files, n),
        }[wl] + "; min_width=100 max_width=1e6 window_width=10000 prior_segments_per_second=10"
        out = {
            "metric": METRIC, "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None,
            "dtype": "int32/f64 (exact integer block sums of fp32 or int16 samples; decisions in fp64)", "data": "synthetic",
            "config": {"workload": workload_text, "name": wl, "samples_per_step": samples_per_step, "boundaries": n_bounds,
                       "segment_stats_in_step": bool(args.stats), "checks": check, "streams": T,
                       # what torch.distributed (RCCL on the GPU box) itself reports, and every rank's own clock
                       "ranks_seen": dist.get_world_size() if use_dist else 1,
                       # every rank's boundary count of its last batch (the gathered counts when N > 1)
                       "per_rank_boundaries": [int(x_) for x_ in n_bounds],
                       "ms_per_step_per_rank": [round(x_, 4) for x_ in per_rank_ms],
                       "streams_note": ("step k runs on context k %% %d (own HIP stream and scratch, one host thread each): "
                                        "independent batches overlap on the GPU; every step is a complete, synchronised "
                                        "ps_segment_batch" % T) if T > 1 else "one batch at a time"},
            "roofline": {"bound": "hbm", "kernel": ("kernel sequence of one ps_segment_batch (blocksum+spine+bridge+stitch+tree+gather)"
                                                    + ("; %d batches in flight: priced on the job's clock (ms_per_step)" % T if T > 1 else ""))
                         if wl != "files" else "whole step incl. the host->HBM copies (PCIe-bound)",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": round(achieved * 1e9 / HBM_PEAK, 5),
                         # HBM bytes per launch (PMC): of one call at a time, and -- the headline configuration -- per call
                         # with the default number of calls in flight, each on its own trace
                         "traffic": (traffic.get("total_in_flight") if T == traffic.get("in_flight_streams") and traffic.get("total_in_flight") else traffic["total"]) if traffic else None,
                         "traffic_measured_in_run": False,       # read from the committed PMC summary of the same command (traffic_source)
                         "traffic_one_call_at_a_time": traffic["total"] if traffic else None,
                         "traffic_source": ((traffic.get("source_in_flight") + "; " if T == traffic.get("in_flight_streams") and traffic.get("total_in_flight") else "") + traffic["source"]) if traffic else None,
                         "traffic_over_algorithmic": round((traffic.get("total_in_flight") if T == traffic.get("in_flight_streams") and traffic.get("total_in_flight") else traffic["total"]) / per_gpu_bytes, 3) if traffic else None,
                         "algorithmic_bytes_per_launch": int(per_gpu_bytes),
                         "longest_kernel": dom.replace("_ms", "_kernel"),
                         "streaming_kernel": {"name": "blocksum_kernel", "ms": round(kern["blocksum_ms"], 4),
                                              "bytes_per_launch": k0_bytes, "bytes_source": k0_bytes_src,
                                              "achieved": round(k0_bytes / (kern["blocksum_ms"] * 1e-3) / 1e9, 1) if kern["blocksum_ms"] > 0 else None,
                                              "frac": round(k0_bytes / (kern["blocksum_ms"] * 1e-3) / HBM_PEAK, 4) if kern["blocksum_ms"] > 0 else None},
                         "traffic_per_kernel": traffic["per_kernel"] if traffic else None,
                         # the call's wave-level vector instructions (SQ_INSTS_VALU of the same PMC passes) at 4 cycles each on
                         # 1 024 SIMDs -- reported beside the roofline (DESIGN.md 6 says what bounds the path)
                         "issue_bound": None if not (traffic and traffic.get("valu_per_kernel")) else {
                             "wave_instructions": int(sum(traffic["valu_per_kernel"].values())),
                             "ms_at_valu_peak": round(sum(traffic["valu_per_kernel"].values()) * 4 / (1024 * 2.4e9) * 1e3, 4),
                             "frac_of_valu_peak": round(sum(traffic["valu_per_kernel"].values()) * 4 / (1024 * 2.4e9) * 1e3 / ms_per_step, 4),
                             "per_kernel": traffic["valu_per_kernel"], "source": traffic.get("valu_source"),
                             "note": "256 CUs x 4 SIMDs, one wave64 vector instruction per 4 cycles at 2.4 GHz; an accounting figure, not the "
                                     "bound: removing 6.8 M of K0's instructions does not move the step (round 6, "
                                     "profiles/r06_experiments/r6_k0_grp_probe.txt; DESIGN.md 6: K0 is bound by HBM, and what it costs the "
                                     "scan kernels is register residency)"},
                         # the secondary bound is instruction issue -- candidate positions the windows of one step
                         # cover (every one is decided: evaluated or excluded by a bound) per second of the job's clock
                         "evaluations_per_s": round(tm["candidates"] * world / (ms_per_step * 1e-3), 1) if wl != "files" else None,
                         "evaluations_per_sample": round(tm["candidates"] / max(1, (trace.numel() if trace is not None else n)), 3),
                         "sequence_ms": round(seq_ms, 4),
                         "single_stream": None if single is None else {
                             "ms_per_step": round(single[0], 4), "sequence_ms": round(single[1], 4),
                             "achieved": round(per_gpu_bytes / (single[1] * 1e-3) / 1e9, 2),
                             "frac": round(per_gpu_bytes / (single[1] * 1e-3) / HBM_PEAK, 5),
                             "note": "one call at a time: sequence_ms = first upload .. last result copy of a call (two HIP events)"},
                         "kernel_ms": {k: round(v, 4) for k, v in kern.items()},
                         "kernel_ms_note": "per-phase HIP events, separate untimed single-stream pass of the same steps (an event "
                                           "between two kernels costs ~6 us of idle GPU, so the timed region records only start "
                                           "and end; with several streams the kernels of different calls overlap and run longer "
                                           "each)"},
            "whole_step_frac_of_hbm_roofline": round(per_gpu_bytes / (ms_per_step * 1e-3) / HBM_PEAK, 5),
            "work": {k: tm[k] for k in ("windows", "windows_spine", "windows_bridge", "windows_tree", "candidates", "tiles", "tree_jobs",
                                       "exact_rescans", "full_exact_scans")},
            # fallbacks of the last step: host-stitch repairs (a seam gave up: BR_MAX anchors), calls redone on the
            # LDS-window path (counts too wide for the block sums), full fp64 window scans
            "host": {"omp_num_threads": os.environ.get("OMP_NUM_THREADS"), "torch_threads": torch.get_num_threads(),
                     "affinity_cpus": len(os.sched_getaffinity(0)), "under_torchrun": "TORCHELASTIC_RUN_ID" in os.environ,
                     "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                     # libporeseg.so and pypore_amd read no PORESEG_* variable (round 6): listed so that a line measured with
                     # --diag-env or a diagnostic library says so itself
                     "poreseg_env_seen": {k_: v_ for k_, v_ in sorted(os.environ.items()) if k_.startswith("PORESEG_")},
                     "poreseg_env_applied": bool(args.diag_env), "library": os.path.basename(_lib.LIB_PATH),
                     "library_version": _lib.lib().ps_version().decode()},
            "fallbacks": {"host_stitch": int(tm["repairs"] >= 1000000), "seam_repairs": int(tm["repairs"] % 1000000),
                          "wide_range_redo": int(tm.get("wide_redo", 0)), "full_exact_scans": int(tm["full_exact_scans"]),
                          "near_ties": int(tm.get("near_ties", 0))},
        }
        # ---- the same job on what a file holds: int16 ADC counts (read_abf.py:208-210) -------------------------
        # BASELINE config 3 beside the fp32 headline: one 1e8-sample .abf-shaped int16 trace per stream,
        # lambda_event_parser(threshold=90) then per-event SpeedyStatSplit, end to end on the GPU, T batches in flight like
        # the headline; priced at ITS algorithmic bytes, 2 B per sample.  Outside the timed region of `value`.
        if wl == "trace" and not args.no_detail and world == 1 and n == 100_000_000 and not args.dwell:
            from pypore_amd import pipeline
            ftraces = []
            for t_ in range(T):
                fe, fl, _ = synth.file_trace_table(n, seed + 1000 * t_)
                ftraces.append(ctx.synth_trace(n, seed + 1000 * t_, fe, fl, dtype=torch.int16))

            def fstep(cx, k, t):
                return pipeline.segment_file_trace(ftraces[t], synth.QUANTUM, params, threshold=90.0, ctx=cx)[2]

            def timed3(k_, job):
                """Three repetitions of k_ pool jobs: (median seconds per job, all three in ms, the last results) -- a side
                measurement of a few milliseconds is at the mercy of one hiccup of the box; the median is not."""
                ts, res = [], None
                for _ in range(3):
                    t1_ = time.perf_counter()
                    res = pool.run(k_, job)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t1_) / k_)
                return sorted(ts)[1], [round(x * 1e3, 4) for x in ts], res

            pool.run(2 * T, fstep)
            torch.cuda.synchronize()
            kf = 40                                      # (its own step count: a side measurement outside the timed region)
            tf, tf_runs, fres = timed3(kf, fstep)

            def fstep2(cx, k, t):                        # the two calls of rounds 3-5 (two passes over the samples), for comparison
                return pipeline.segment_file_trace(ftraces[t], synth.QUANTUM, params, threshold=90.0, ctx=cx, single_pass=False)[2]

            pool.run(2 * T, fstep2)
            torch.cuda.synchronize()
            tf2, tf2_runs, fres2 = timed3(kf, fstep2)
            same_two = bool(torch.equal(fres[-1], fres2[-1]))
            out["int16_file"] = {
                "workload": "BASELINE config 3: one %.0e-sample int16 .abf-shaped trace per stream @100 kHz, lambda_event_parser("
                            "threshold=90) -> per-event SpeedyStatSplit, end to end on the GPU, %d batches in flight" % (n, T),
                "ms_per_step": round(tf * 1e3, 4), "value": round(n / tf / 1e6, 2), "unit": "Msamples/s", "steps": kf,
                "repetitions_ms": tf_runs, "how": "median of three repetitions of %d steps" % kf,
                "boundaries": int(fres[-1].numel()),
                "route": "ps_detect_segment_trace: one pass over the samples (K0 over the whole trace serves the detector and every event)",
                "two_calls_ms_per_step": round(tf2 * 1e3, 4), "two_calls_repetitions_ms": tf2_runs, "two_calls_same_boundaries": same_two,
                "roofline": {"bound": "hbm", "algorithmic_bytes_per_launch": 2 * n, "achieved": round(2 * n / tf / 1e9, 2),
                             "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(2 * n / tf / HBM_PEAK, 5)}}
            del ftraces
        # ---- BASELINE config 2: 1 024 events x 50 000 samples in ONE call -----------------------
        if wl == "trace" and not args.no_detail and world == 1 and n == 100_000_000 and not args.dwell:
            n_ev2, ln2 = 1024, 50000
            e2, l2 = [], []
            for e_ in range(n_ev2):
                for k_ in range(5):
                    e2.append(e_ * ln2 + (k_ + 1) * 10000)
                    l2.append(int(synth.LEVEL_COUNTS[k_]))
            t2_ = ctx.synth_trace(n_ev2 * ln2, 7, np.array(e2), np.array(l2, dtype=np.int32), dtype=torch.float32)
            off2 = np.arange(n_ev2 + 1, dtype=np.int64) * ln2
            for _ in range(3):
                b2, o2, _ = ctx.segment_batch(t2_, off2, params, synth.QUANTUM, want_stats=False)
            torch.cuda.synchronize()
            c2_runs = []
            for _ in range(3):                           # (median of three repetitions of 20 calls, like timed3)
                t1 = time.perf_counter()
                s2 = 0.0
                for _ in range(20):
                    b2, o2, _ = ctx.segment_batch(t2_, off2, params, synth.QUANTUM, want_stats=False)
                    s2 += ctx.seq_ms()
                torch.cuda.synchronize()
                c2_runs.append(((time.perf_counter() - t1) / 20, s2))
            tc2, s2 = sorted(c2_runs)[1]
            # the same batch shape with T batches in flight, each on its own events (noise seeds differ), like the headline
            # and `int16_file`: what a caller with more than one batch gets out of the pool
            t2s = [t2_] + [ctx.synth_trace(n_ev2 * ln2, 7 + 13 * t_, np.array(e2), np.array(l2, dtype=np.int32), dtype=torch.float32)
                           for t_ in range(1, T)]

            def c2step(cx, k, t):
                return cx.segment_batch(t2s[t], off2, params, synth.QUANTUM, want_stats=False)[1]

            pool.run(2 * T, c2step)
            torch.cuda.synchronize()
            k2 = 48
            tp2_runs, o2s = [], None
            for _ in range(3):
                t1 = time.perf_counter()
                o2s = pool.run(k2, c2step)
                torch.cuda.synchronize()
                tp2_runs.append((time.perf_counter() - t1) / k2)
            tp2 = sorted(tp2_runs)[1]
            out["config2"] = {
                "workload": "BASELINE config 2: %d events x %d samples (5 levels x 10 000 each), one ps_segment_batch, one call at a time" % (n_ev2, ln2),
                "ms_per_step": round(tc2 * 1e3, 4), "sequence_ms": round(s2 / 20, 4), "value": round(n_ev2 * ln2 / tc2 / 1e6, 2),
                "unit": "Msamples/s", "steps": 20, "repetitions_ms": [round(x[0] * 1e3, 4) for x in c2_runs],
                "how": "median of three repetitions of 20 calls", "boundaries": int(b2.numel()),
                "events_with_exactly_their_four_steps": int(np.sum(np.diff(o2) == 4)),
                "roofline": {"bound": "hbm", "algorithmic_bytes_per_launch": 4 * n_ev2 * ln2, "achieved": round(4 * n_ev2 * ln2 / tc2 / 1e9, 2),
                             "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(4 * n_ev2 * ln2 / tc2 / HBM_PEAK, 5)},
                "in_flight": {"batches_in_flight": T, "steps": k2, "ms_per_step": round(tp2 * 1e3, 4),
                              "repetitions_ms": [round(x * 1e3, 4) for x in tp2_runs],
                              "value": round(n_ev2 * ln2 / tp2 / 1e6, 2), "unit": "Msamples/s",
                              "frac": round(4 * n_ev2 * ln2 / tp2 / HBM_PEAK, 5),
                              "events_with_exactly_their_four_steps_min": int(min(np.sum(np.diff(o_) == 4) for o_ in o2s))}}
            del t2_, t2s
        # ---- PCIe-inclusive rate (report H2D-inclusive separately; never `value`) ------------------
        if wl == "trace" and not args.no_h2d and world == 1:
            P = 8
            pr = pdist.shard_ranges(n, P, 8 * W)
            plen = max(hi_ - lo_ for lo_, hi_ in pr)
            hbuf = torch.empty(n, dtype=torch.float32, pin_memory=True)
            hbuf.copy_(trace)
            torch.cuda.synchronize()
            dbuf = [torch.empty(plen, dtype=torch.float32, device=device) for _ in range(2)]
            cs = torch.cuda.Stream(device=device)

            def h2d_step():
                evs = [torch.cuda.Event() for _ in pr]
                with torch.cuda.stream(cs):
                    dbuf[0][:pr[0][1] - pr[0][0]].copy_(hbuf[pr[0][0]:pr[0][1]], non_blocking=True)
                    evs[0].record(cs)
                pieces = []
                for j, (plo, phi) in enumerate(pr):
                    evs[j].synchronize()
                    if j + 1 < P:
                        with torch.cuda.stream(cs):
                            dbuf[(j + 1) % 2][:pr[j + 1][1] - pr[j + 1][0]].copy_(hbuf[pr[j + 1][0]:pr[j + 1][1]], non_blocking=True)
                            evs[j + 1].record(cs)
                    b, o, st, sp = ctx.segment_batch(dbuf[j % 2][:phi - plo], np.array([0, phi - plo], dtype=np.int64), params,
                                                     synth.QUANTUM, want_stats=False, want_spine=True)
                    pieces.append((plo, phi, b.cpu().numpy(), sp.cpu().numpy()))
                return pdist.stitch_pieces(pieces, n, W, mw)

            for _ in range(2):
                hb = h2d_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                hb = h2d_step()
            torch.cuda.synchronize()
            th = (time.perf_counter() - t1) / reps
            out["h2d_inclusive"] = {"value": round(n / th / 1e6, 1), "unit": "Msamples/s", "ms_per_trace": round(th * 1e3, 3),
                                    "how": "trace in pinned host memory, %d pieces with an 8 W halo, piece k+1 copied on a second "
                                           "stream while piece k is segmented, pieces joined at common spine anchors" % P,
                                    "pcie_GBps": round(4 * n / th / 1e9, 1),
                                    "boundaries_equal_resident_run": bool(np.array_equal(hb, bounds.cpu().numpy()))}
            del hbuf, dbuf
            # the same trace as int16 ADC counts -- what a file holds (read_abf.py:208-210): half the bytes over PCIe
            d16 = synth.dwell_table(seed, n, *args.dwell) if args.dwell else synth.dwell_table(seed, n)
            t16 = ctx.synth_trace(n, seed, np.cumsum(d16), synth.LEVEL_COUNTS[np.arange(len(d16)) % 5].astype(np.int32), dtype=torch.int16)
            hbuf = torch.empty(n, dtype=torch.int16, pin_memory=True)
            hbuf.copy_(t16)
            del t16
            torch.cuda.synchronize()
            dbuf = [torch.empty(plen, dtype=torch.int16, device=device) for _ in range(2)]
            for _ in range(2):
                hb16 = h2d_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                hb16 = h2d_step()
            torch.cuda.synchronize()
            th16 = (time.perf_counter() - t1) / reps
            out["h2d_inclusive"]["int16"] = {"value": round(n / th16 / 1e6, 1), "unit": "Msamples/s", "ms_per_trace": round(th16 * 1e3, 3),
                                             "pcie_GBps": round(2 * n / th16 / 1e9, 1),
                                             "boundaries_equal_resident_run": bool(np.array_equal(hb16, bounds.cpu().numpy()))}
            del hbuf, dbuf
        # ---- CPU baseline: the oracle on this host, one thread and one thread per core ------------------------
        if not args.no_cpu and wl in ("trace", "file"):
            import oracle
            m = min(n, args.cpu_samples)
            x = trace[:m].cpu().numpy().astype(np.float64)
            if wl == "file":
                x = x * synth.QUANTUM
            cores = os.cpu_count() or 1
            t1 = time.perf_counter()
            if wl == "file":
                es, el = oracle.lambda_events(x, threshold=90.0)
                refs = [oracle.parse(x[a:a + l], **PARAMS) for a, l in zip(es, el)]
                ref = np.concatenate(refs) if refs else np.zeros(0, np.int32)
            else:
                ref = oracle.parse(x, **PARAMS)
            t2 = time.perf_counter()
            got = bounds.cpu().numpy()
            # prefix property: parse(x[:m]) agrees with parse(x) away from the cut
            kcut = int(np.searchsorted(ref, m - 4 * W)) if wl == "trace" else int(sum(len(r) for r in refs[:-1]))
            out["cpu_baseline"] = {"value": round(m / (t2 - t1) / 1e6, 3), "unit": "Msamples/s", "cores": 1,
                                   "kind": "port", "cpu_model": cpu_model(), "host_cores": cores,
                                   "sample": "first %d samples of rank 0's trace, oracle/statsplit_oracle.c "
                                             "(gcc -O2), single thread" % m,
                                   "prefix_boundaries_equal": bool(np.array_equal(ref[:kcut], got[:kcut]))}
            if wl == "trace" and cores > 1:
                t3 = time.perf_counter()
                refp = cpu_parse_tiled(x, cores, 8 * W)
                t4 = time.perf_counter()
                out["cpu_baseline"]["all_cores"] = {"value": round(m / (t4 - t3) / 1e6, 3), "unit": "Msamples/s", "cores": cores,
                                                    "how": "one thread per host core over contiguous tiles with an 8 W halo, "
                                                           "joined at common spine anchors (same samples)",
                                                    "equal_single_thread": bool(np.array_equal(refp, ref))}
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
