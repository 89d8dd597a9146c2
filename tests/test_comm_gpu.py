"""The C ABI's multi-GPU entry points (include/poreseg_comm.h, libporeseg_comm.so) on ONE GPU: a communicator of one rank
made both ways (ps_comm_init_rank, ps_comm_init_all), the fixed-shape boundary gather, and dist.BoundaryGather on the
library's gather instead of torch.distributed's.  More than one rank needs more than one GPU: the same calls, world > 1
(the gloo tests cover the sharding logic around them).  Runs in a child process: it initialises torch.distributed."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
import torch.distributed as dist
from pypore_amd import _comm, _lib, engine, synth
from pypore_amd import dist as pdist
torch.cuda.set_device(0)
cap = 4096
send = torch.zeros(cap, dtype=torch.int32, device="cuda")
send[0] = 5
send[_comm.HEADER:_comm.HEADER + 5] = torch.tensor([3, 14, 15, 92, 65], dtype=torch.int32, device="cuda")
for how in ("rank", "all"):
    c = _comm.Comm.for_rank(1, 0, _comm.unique_id(), 0) if how == "rank" else _comm.Comm.all(1)[0]
    assert (c.world, c.rank) == (1, 0)
    recv = torch.full((cap,), -1, dtype=torch.int32, device="cuda")
    c.gather_bounds(send, recv)
    torch.cuda.synchronize()
    assert torch.equal(recv, send), how
    c.close()
# the grouped form of ps_comm_init_all's communicators
import ctypes
cs = _comm.Comm.all(1)
recv = torch.full((cap,), -1, dtype=torch.int32, device="cuda")
hs = (ctypes.c_void_p * 1)(cs[0].handle)
sp = (ctypes.c_void_p * 1)(send.data_ptr()); rp = (ctypes.c_void_p * 1)(recv.data_ptr())
st = (ctypes.c_void_p * 1)(torch.cuda.current_stream().cuda_stream)
_comm.check(_comm.lib().ps_gather_bounds_all(hs, 1, sp, rp, cap, st))
torch.cuda.synchronize()
assert torch.equal(recv, send)
cs[0].close()
# dist.BoundaryGather on the library's gather: the segmenter's own output buffer goes out as it is
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
ctx = engine.context(0)
n = 2_000_000
d = synth.dwell_table(11, n); lv = synth.LEVEL_COUNTS[np.arange(len(d)) %% 5].astype(np.int32)
t = ctx.synth_trace(n, 11, np.cumsum(d), lv, dtype=torch.float32)
p = _lib.split_params(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
res = {}
for backend in ("torch", "library"):
    g = pdist.BoundaryGather(1 << 16, "cuda:0", backend=backend)
    b, _, _ = ctx.segment_batch(t, np.array([0, n], dtype=np.int64), p, synth.QUANTUM, want_stats=False, lead=pdist.BoundaryGather.HEADER)
    parts = g.result(g.submit(b))
    assert len(parts) == 1 and torch.equal(parts[0], b) and b.numel() > 100
    res[backend] = parts[0].clone()
assert torch.equal(res["torch"], res["library"])
dist.destroy_process_group()
print("comm ok")
"""


def test_comm_library_world_of_one():
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "comm ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


CHILD2 = r"""
import ctypes, os, sys
sys.path.insert(0, %r)
import torch
from pypore_amd import _comm
n = torch.cuda.device_count()
assert n >= 2
cap = 1024
cs = _comm.Comm.all(2)                                   # ps_comm_init_all: one process drives both GPUs (ncclCommInitAll)
assert [(c.world, c.rank) for c in cs] == [(2, 0), (2, 1)]
send, recv = [], []
for r in range(2):
    torch.cuda.set_device(r)
    s = torch.zeros(cap, dtype=torch.int32, device="cuda:%%d" %% r)
    s[0] = 3 + r
    s[_comm.HEADER:_comm.HEADER + 3 + r] = torch.arange(3 + r, dtype=torch.int32, device=s.device) * (r + 1) + 7
    send.append(s); recv.append(torch.full((2 * cap,), -1, dtype=torch.int32, device=s.device))
hs = (ctypes.c_void_p * 2)(*[c.handle for c in cs])
sp = (ctypes.c_void_p * 2)(*[t.data_ptr() for t in send]); rp = (ctypes.c_void_p * 2)(*[t.data_ptr() for t in recv])
st = (ctypes.c_void_p * 2)(*[torch.cuda.current_stream(r).cuda_stream for r in range(2)])
_comm.check(_comm.lib().ps_gather_bounds_all(hs, 2, sp, rp, cap, st))
for r in range(2):
    torch.cuda.synchronize(r)
    rows = recv[r].view(2, cap).cpu()
    for q in range(2):                                   # every rank holds every rank's slot: count in element 0, payload from HEADER on
        assert int(rows[q, 0]) == 3 + q and torch.equal(rows[q, _comm.HEADER:_comm.HEADER + 3 + q], send[q][_comm.HEADER:_comm.HEADER + 3 + q].cpu())
for c in cs:
    c.close()
print("comm2 ok")
"""


def test_comm_library_two_ranks_in_one_process():
    """the library's gather with MORE than one rank -- ps_comm_init_all over two GPUs driven by one process, the
    grouped all-gather, every rank ends up with every rank's slot.  Needs two GPUs (skipped on the one-GPU boxes the builder
    has; the driver's multi-chip run is where it executes)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, "-c", CHILD2 % ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "comm2 ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
