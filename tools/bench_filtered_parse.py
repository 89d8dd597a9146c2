import sys, time, json, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
import torch
from pypore_amd import synth, engine, _lib
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
ctx = engine.context(0)
import os
if os.environ.get('TILE'): ctx.set_tiling(int(os.environ['TILE']), 0)
# a filtered, centred, finely quantised event as Event.parse builds it: 1e6 samples
k = synth.random_dwell_counts(1_000_000, 5, 1000, 20000)
dev = torch.from_numpy(k.astype(np.int16)).cuda()
y = ctx.filter_bessel(dev, synth.QUANTUM).cpu().numpy()
c0 = y.mean(); span = np.abs(y - c0).max(); fq = 2.0 ** (int(np.ceil(np.log2(span * 1.01))) - 22)
z = (np.rint((y - np.rint(c0 / fq) * fq) / fq) * fq).astype(np.float32)
t = torch.from_numpy(z).cuda()
p = _lib.split_params(prior_segments_per_second=10.)
off = np.array([0, len(z)], dtype=np.int64)
for _ in range(2):
    b, boff, _ = ctx.segment_batch(t, off, p, fq, want_stats=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    b, boff, _ = ctx.segment_batch(t, off, p, fq, want_stats=False)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
ctx.set_option("timing", 2); ctx.segment_batch(t, off, p, fq, want_stats=False); tm = ctx.timings(); ctx.set_option("timing", 1)
print({k: round(v, 3) for k, v in tm.items() if k.endswith("_ms")})
print("filtered 1e6-sample event on the 2^%d grid: %.3f ms, %d boundaries" % (int(np.log2(fq)), ms, b.numel()), ctx.timings())

# the same for a file's worth of events in ONE call (what File.parse / parse_batch hand over): E filtered events of 10^6
# samples each, all on the grid of the widest one
E = int(os.environ.get("BATCH", "32"))
zs = []
for e in range(E):
    ke = synth.random_dwell_counts(1_000_000, 100 + e, 1000, 20000)
    ye = ctx.filter_bessel(torch.from_numpy(ke.astype(np.int16)).cuda(), synth.QUANTUM).cpu().numpy()
    zs.append(ye - ye.mean())
span = max(np.abs(v).max() for v in zs); fq = 2.0 ** (int(np.ceil(np.log2(span * 1.01))) - 22)
tb = torch.from_numpy(np.concatenate([(np.rint(v / fq) * fq).astype(np.float32) for v in zs])).cuda()
offb = np.arange(E + 1, dtype=np.int64) * 1_000_000
for wide in (1, 0):
    ctx.set_option("wide_bs", wide)
    for _ in range(2):
        b, boff, _ = ctx.segment_batch(tb, offb, p, fq, want_stats=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        b, boff, _ = ctx.segment_batch(tb, offb, p, fq, want_stats=False)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
    ctx.set_option("timing", 2); ctx.segment_batch(tb, offb, p, fq, want_stats=False); tm = ctx.timings(); ctx.set_option("timing", 1)
    print("%d filtered events in one call, %s: %.3f ms = %.3f ms per event (%.2f Gsamples/s), %d boundaries, route %d" %
          (E, "64-bit digest" if wide else "LDS-window kernels", ms, ms / E, E * 1e6 / ms / 1e6, b.numel(), tm["wide_redo"]),
          {k: round(v, 3) for k, v in tm.items() if k.endswith("_ms")})
ctx.set_option("wide_bs", 1)
