import sys, time, json, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
import torch
from pypore_amd import synth, engine, _lib
ctx = engine.context(0)
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
