"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm, "gloo"
on CPU for tests).  The path shards embarrassingly -- events / files are independent units
(Experiment.parse iterates files, then events: DataTypes.py:968-984) -- so there is NO data-path
collective; the only exchange is the final gather of boundary indices:
one all_gather of the per-rank counts, then one padded all_gather of the int32 payload
(tens of KB: latency-bound, a single step over the xGMI links).
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
