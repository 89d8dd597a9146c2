"""Where the host time of one segment_batch call goes: Python wrapper vs the C call vs the device work."""
import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pypore_amd import _lib, engine, synth
ctx = engine.context(0)
n = 10**8
d = synth.dwell_table(2024, n); ends = np.cumsum(d)
lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
trace = ctx.synth_trace(n, 2024, ends, lv, dtype=torch.float32)
ev_off = np.array([0, n], dtype=np.int64)
params = _lib.split_params(**bench.PARAMS)
out = torch.empty(n // 100 + 1, dtype=torch.int32, device=trace.device)
for _ in range(50):
    ctx.segment_batch(trace, ev_off, params, synth.QUANTUM, want_stats=False, out=out)
K = 300
t0 = time.perf_counter()
for _ in range(K):
    ctx.segment_batch(trace, ev_off, params, synth.QUANTUM, want_stats=False, out=out)
t_py = (time.perf_counter() - t0) / K
# the bare C call with prebuilt arguments
fmt = _lib.SampleFormat(_lib.PS_DTYPE_F32, 0, float(synth.QUANTUM))
off_p = ev_off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
boff = np.zeros(2, dtype=np.int64); boff_p = boff.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
L = ctx.L; h = ctx.handle; sp = ctypes.c_void_p(trace.data_ptr()); op = ctypes.c_void_p(out.data_ptr()); cap = out.numel()
t0 = time.perf_counter(); seq = 0.0
for _ in range(K):
    L.ps_segment_batch_ex(h, sp, ctypes.byref(fmt), off_p, 1, ctypes.byref(params), op, cap, boff_p, None, None)
t_c = (time.perf_counter() - t0) / K
seq = ctx.timings()["seq_ms"]
print("python wrapper %.1f us/call, bare C call %.1f us/call, device sequence %.1f us" % (t_py * 1e6, t_c * 1e6, seq * 1e3))
