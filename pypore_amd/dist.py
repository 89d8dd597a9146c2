"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm, "gloo"
on CPU for tests).  The path shards embarrassingly -- events / files are independent units
(Experiment.parse iterates files, then events: DataTypes.py:968-984) -- so there is NO data-path
collective; the only exchange is the final gather of boundary indices, tens of KB
and latency-bound: BoundaryGather does it as one fixed-shape all_gather per batch, enqueued without
a host sync and overlapped with the next batch; gather_varlen (counts, then padded payload) is the
general two-collective form.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_units(lengths, world_size):
    """Greedy longest-first partition of independent units (events or files) by sample count.
    Returns a list (per rank) of unit indices; deterministic, identical on every rank."""
    lengths = np.asarray(lengths, dtype=np.int64)
    order = np.argsort(-lengths, kind="stable")
    loads = np.zeros(world_size, dtype=np.int64)
    shards = [[] for _ in range(world_size)]
    for u in order:
        r = int(np.argmin(loads))
        shards[r].append(int(u))
        loads[r] += lengths[u]
    return [sorted(s) for s in shards]


def gather_varlen(local, group=None):
    """All-gathers 1-D tensors of different lengths (same dtype/device on every rank).
    Returns the list of per-rank tensors.  Two collectives: counts, then padded payload."""
    world = dist.get_world_size(group)
    n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(max(counts), 1)
    pad = torch.zeros(m, dtype=local.dtype, device=local.device)
    pad[:local.numel()] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return [o[:c] for o, c in zip(out, counts)]


class _StreamDone(object):
    """wait() of a gather queued on a stream by the library (BoundaryGather backend "library"): an event on that stream."""

    def __init__(self, stream):
        self.ev = None
        if stream is not None:                           # (None: a transport that completed synchronously -- CPU tensors in the self-test)
            self.ev = torch.cuda.Event()
            self.ev.record(stream)

    def wait(self):
        if self.ev is not None:
            torch.cuda.current_stream().wait_event(self.ev)


class BoundaryGather:
    """The boundary gather as ONE fixed-shape collective per batch and no host synchronisation on the
    submitting side: every rank contributes a slot of `capacity` elements, element 0 = its count,
    the payload from element HEADER on.  submit() enqueues the all_gather asynchronously (RCCL's own stream, so
    the next batch's kernels overlap it); result() waits, reads the counts and slices.  A contribution that
    does not fit is seen by EVERY rank in the gathered counts, so all ranks fall back to gather_varlen
    for that batch together (no extra agreement round).  `depth` batches may be in flight.

    No copy when the boundaries already sit HEADER elements into a buffer of at least `capacity` elements
    (engine.segment_batch(..., lead=BoundaryGather.HEADER)): the count goes into the buffer's first element and the
    buffer's head is sent as it is (whatever lies behind the count's worth of payload is ignored by the receiver)."""
    HEADER = 4          # elements before the payload (16 bytes: the payload keeps its alignment)

    def __init__(self, capacity, device, dtype=torch.int32, group=None, depth=2, backend="torch", comm=None):
        """backend "torch" (default): torch.distributed's all_gather on the group (RCCL on GPUs, gloo in the CPU tests);
        "library": the C ABI's own entry point, ps_gather_bounds of libporeseg_comm.so (include/poreseg_comm.h) -- the
        same ncclAllGather underneath, on a communicator the library sets up (its id travels over the torch group once).
        comm: (library backend) an object with ps_gather_bounds' contract -- gather_bounds(send[capacity], recv[world *
        capacity]), every rank's slot at rank * capacity -- to use instead of a new RCCL communicator: bench.py's
        --selftest-dist passes one that moves CPU tensors over gloo, so that the slot layout of this backend (header,
        in-place send, overflow and the fall-back all ranks take together) is exercised with 8 ranks and no GPU."""
        self.group = group
        self.world = dist.get_world_size(group)
        self.cap = int(capacity)
        self.comm = None
        self._own_comm = False
        if backend == "library":
            assert dtype == torch.int32, "ps_gather_bounds gathers int32 boundaries"
            if comm is not None:
                self.comm = comm
            else:
                from . import _comm
                ids = [_comm.unique_id() if dist.get_rank(group) == 0 else None]
                dist.broadcast_object_list(ids, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                dev = torch.device(device)
                self.comm = _comm.Comm.for_rank(self.world, dist.get_rank(group), ids[0], dev.index if dev.index is not None else torch.cuda.current_device())
                self._own_comm = True
        else:
            assert backend == "torch", backend
        self.slots = [dict(send=torch.zeros(self.cap, dtype=dtype, device=device),
                           recv=torch.zeros(self.world * self.cap, dtype=dtype, device=device),
                           work=None, local=None) for _ in range(depth)]
        self.k = 0

    def submit(self, local):
        """Enqueues the gather of `local` (1-D, the gather's dtype/device); returns a ticket."""
        s = self.slots[self.k % len(self.slots)]
        assert s["work"] is None, "BoundaryGather: result() of an earlier batch is outstanding"
        n = local.numel()
        base = local._base
        if (base is not None and base.dim() == 1 and local.storage_offset() == self.HEADER and base.numel() >= self.cap
                and base.dtype == s["send"].dtype):
            send = base[:self.cap]                       # in place: header + payload are already laid out
            send[:1].fill_(n)
        else:
            send = s["send"]
            send[:1].fill_(n)
            if n <= self.cap - self.HEADER:
                send[self.HEADER:self.HEADER + n].copy_(local)
        s["local"] = local
        if self.comm is not None:
            self.comm.gather_bounds(send, s["recv"])         # on the current stream: ordered behind the kernels that wrote `send`
            s["work"] = _StreamDone(torch.cuda.current_stream(send.device) if send.is_cuda else None)
        else:
            s["work"] = dist.all_gather_into_tensor(s["recv"], send, group=self.group, async_op=True)
        self.k += 1
        return self.k - 1

    def close(self):
        """Gives the library backend's communicator back (ps_comm_destroy); the torch backend has nothing to release."""
        if self.comm is not None and self._own_comm:
            self.comm.close()
        self.comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:                                # noqa: BLE001 -- interpreter shutdown
            pass

    def result(self, ticket, host=True):
        """Per-rank tensors of batch `ticket` (views into the slot: consume before `depth` more submits).
        host=False: only waits for the collective (no host sync) and returns the gathered rows [world, capacity] on
        the device, count in column 0 -- for consumers that stay on the device; an overflowing contribution then
        shows as a count above capacity - HEADER."""
        s = self.slots[ticket % len(self.slots)]
        s["work"].wait()
        s["work"] = None
        rows = s["recv"].view(self.world, self.cap)
        local, s["local"] = s["local"], None
        if not host:
            return rows
        counts = [int(c) for c in rows[:, 0].cpu().tolist()]
        if max(counts) > self.cap - self.HEADER:
            return gather_varlen(local, self.group)
        return [rows[r, self.HEADER:self.HEADER + c] for r, c in enumerate(counts)]


def segment_units_sharded(unit_lengths, segment_fn, device=None, group=None):
    """Segments independent units across the ranks of `group` and gathers the boundaries.

    unit_lengths  sample count of every unit (known on every rank)
    segment_fn    callable(list_of_unit_indices) -> list of int32 numpy arrays, one per unit:
                  the local segmenter (SpeedyStatSplit.parse_batch on this rank's GPU)
    Returns, on every rank, a list with the boundary array of every unit (global unit order).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    shards = shard_units(unit_lengths, world)
    mine = shards[rank]
    local = segment_fn(mine) if mine else []
    assert len(local) == len(mine)
    dev = device if device is not None else torch.device("cpu")
    counts = torch.tensor([len(b) for b in local], dtype=torch.int32, device=dev)
    payload = torch.from_numpy(np.concatenate([np.asarray(b, dtype=np.int32) for b in local])
                               if local else np.zeros(0, np.int32)).to(dev)
    all_counts = gather_varlen(counts, group)
    all_payload = gather_varlen(payload, group)
    out = [None] * len(unit_lengths)
    for r in range(world):
        c = all_counts[r].cpu().numpy()
        p = all_payload[r].cpu().numpy()
        offs = np.concatenate(([0], np.cumsum(c)))
        for k, u in enumerate(shards[r]):
            out[u] = p[offs[k]:offs[k + 1]].copy()
    return out


# ---- one long trace sharded across GPUs (BASELINE config 5) ---------------------------------------
# rec(a, N) depends only on (a, N, data): two segmentations of overlapping pieces of one trace are
# identical after any common SPINE anchor (a breakpoint of the top-level chain of right recursions),
# provided the windows that found it did not touch the end of the piece.  So every rank segments
# its piece [S_r, S_{r+1} + halo) as if it were a whole trace (no sample exchange between GPUs), and
# the pieces are joined at the first spine anchor that rank r (trusted part) and rank r+1 share.

def shard_ranges(n, world_size, halo):
    """Piece of every rank: (lo, hi) with hi = min(n, next shard start + halo)."""
    starts = [(r * n) // world_size for r in range(world_size + 1)]
    return [(starts[r], min(n, starts[r + 1] + (halo if r < world_size - 1 else 0))) for r in range(world_size)]


def stitch_pieces(pieces, n, window_width, min_width, repair=None, halo=None):
    """pieces: per rank (lo, hi, bounds_local int32, is_spine uint8), rank order.  Returns the global
    breakpoint array.

    Two consecutive pieces are joined at the first spine anchor both found, the upstream one by windows that cannot
    have touched its end.  A seam without such an anchor inside the halo (dwells longer than the halo, a flat stretch)
    is REPAIRED when `repair` is given ("extend and re-run that seam only"): rec(a, N) depends only on
    (a, N, data) (cparsers.pyx:180-203), so the chain is re-run from the last trusted upstream anchor a0 over
    [a0, next piece start + 2 halo), `repair(r, lo, hi) -> (bounds_local, is_spine)` segmenting [lo, hi) as a stand-alone
    trace on behalf of upstream piece r; the extension doubles until it shares an anchor with a downstream piece (pieces
    it covers entirely are dropped) or reaches the end of the trace.  Without `repair` such a seam raises RuntimeError.
    CONTRACT of `repair`: [lo, hi) may reach far beyond piece r's own shard plus halo (up to the end of the trace), and the
    result is taken to cover all of it.  A callback that may come up short returns a third value, the number of samples it
    really segmented: the repaired piece then ends there (its last anchors are not trusted), and RuntimeError is raised if
    that does not get past the old end -- a silently truncated piece would otherwise give wrong boundaries without an error."""
    if halo is None:
        halo = 8 * window_width
    out = []
    enter = -1                                   # global position after which the current piece is valid
    cur = None                                   # the upstream piece as (lo, hi, global bounds, spine flags)
    r = 0
    nxt = 0                                      # index of the next piece to take from `pieces`
    n_pieces = len(pieces)
    repairs = 0
    while True:
        if cur is None:
            lo, hi, b, f = pieces[nxt]
            cur = (lo, hi, np.asarray(b, dtype=np.int64) + lo, np.asarray(f, dtype=bool))
            r = nxt
            nxt += 1
        lo, hi, g, f = cur
        if nxt >= n_pieces or hi >= n and nxt >= n_pieces:
            out.append(g[g > enter])
            break
        if hi >= n:                              # the (repaired) upstream piece runs to the end of the trace: done
            out.append(g[g > enter])
            break
        nlo, nhi, nb, nf = pieces[nxt]
        if nhi <= hi and nhi < n:                # downstream piece lies inside the upstream one: nothing new in it
            nxt += 1
            continue
        ng = np.asarray(nb, dtype=np.int64) + nlo
        nspine = set(ng[np.asarray(nf, dtype=bool)].tolist())
        # spine anchors of this piece found by windows that cannot have touched its end
        trust_limit = hi - 2 * window_width - 2 * min_width
        trusted = f & (g > enter) & (g <= trust_limit)
        cand = g[trusted & (g >= nlo)]
        join = next((int(a) for a in cand if int(a) in nspine), None)
        if join is not None:
            out.append(g[(g > enter) & (g <= join)])
            enter = join
            cur = None
            continue
        if repair is None:
            raise RuntimeError("sharded trace: pieces %d and %d share no spine anchor inside the halo "
                               "(increase halo)" % (r, nxt))
        # no common anchor: re-run the chain from the last trusted upstream anchor over a longer stretch
        ta = g[trusted]
        a0 = int(ta[-1]) if ta.size else (enter if enter >= 0 else lo)
        ext = 2 * halo * (1 << min(repairs, 20))
        new_hi = int(min(n, max(hi, nlo) + ext))
        out.append(g[(g > enter) & (g <= a0)])
        enter = max(enter, a0)
        res = repair(r, a0, new_hi)
        rb, rf = res[0], res[1]
        if len(res) > 2 and res[2] is not None:
            # the segmenter says how many samples it really had (a callback that can only reach part of the stretch, e.g.
            # one that slices a local buffer): the piece ends THERE -- anchors next to that end are not trusted -- and a
            # stretch that does not get past the old end cannot repair anything
            got_hi = a0 + int(res[2])
            if got_hi > new_hi or (got_hi < new_hi and got_hi <= max(hi, nlo)):
                raise RuntimeError("sharded trace: the repair of the seam between pieces %d and %d asked for [%d, %d) and "
                                   "got %d samples: the upstream segmenter cannot extend (it must serve any range inside "
                                   "[0, n))" % (r, nxt, a0, new_hi, int(res[2])))
            new_hi = got_hi
        cur = (a0, new_hi, np.asarray(rb, dtype=np.int64) + a0, np.asarray(rf, dtype=bool))
        repairs += 1
    stitch_pieces.last_repairs = repairs
    return np.concatenate(out).astype(np.int64) if out else np.zeros(0, np.int64)


stitch_pieces.last_repairs = 0


def segment_trace_sharded(n, segment_piece_fn, window_width, min_width, halo=None, device=None, group=None):
    """Segments ONE trace of n samples across the ranks of `group`.

    segment_piece_fn(lo, hi) -> (bounds_local int32, is_spine uint8[, n_segmented]): the local segmenter applied to
    samples [lo, hi) as a stand-alone trace (ps_segment_batch_ex with d_is_spine on this rank's GPU).  It is called once
    with this rank's shard plus halo and, when a seam needs repair, with stretches that may lie ANYWHERE inside [0, n) --
    far beyond the shard: it must be able to produce any range (regenerate, re-read from the file), or return as a third
    value how many samples of [lo, hi) it really segmented (stitch_pieces then ends the piece there or raises).
    Every rank returns the full global breakpoint array.  Collectives: the boundary gather, and for every seam that
    finds no common anchor inside the halo one more gather of the stretch the UPSTREAM rank re-segments (all ranks
    walk the same gathered data, so they agree on which seam failed without an extra round)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if halo is None:
        halo = 8 * window_width
    ranges = shard_ranges(n, world, halo)
    lo, hi = ranges[rank]
    b, f = segment_piece_fn(lo, hi)[:2]
    dev = device if device is not None else torch.device("cpu")

    def gather2(bb, ff):
        allb = gather_varlen(torch.from_numpy(np.ascontiguousarray(bb, dtype=np.int32)).to(dev), group)
        allf = gather_varlen(torch.from_numpy(np.ascontiguousarray(ff, dtype=np.uint8)).to(dev), group)
        return allb, allf

    allb, allf = gather2(b, f)
    pieces = [(ranges[r][0], ranges[r][1], allb[r].cpu().numpy(), allf[r].cpu().numpy()) for r in range(world)]

    def repair(r_up, lo2, hi2):
        # the upstream rank re-segments [lo2, hi2); everybody takes part in the gather of that one stretch
        if rank == r_up:
            res = segment_piece_fn(lo2, hi2)
            rb, rf = res[0], res[1]
            ln = np.array([res[2] if len(res) > 2 and res[2] is not None else hi2 - lo2], dtype=np.int64)
        else:
            rb, rf, ln = np.zeros(0, np.int32), np.zeros(0, np.uint8), np.zeros(0, np.int64)
        gb, gf = gather2(rb, rf)
        gl = gather_varlen(torch.from_numpy(ln).to(dev), group)
        return gb[r_up].cpu().numpy(), gf[r_up].cpu().numpy(), int(gl[r_up][0])

    return stitch_pieces(pieces, n, window_width, min_width, repair=repair, halo=halo)
