"""Aligner throughput (GPU box): ps_align_batch on batches of random-walk sequences against one model, next to the
oracle (the C restatement of cSegmentAligner, one core) on a sample of the same batch."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
from pypore_amd import engine
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none

ctx = engine.context(0)
dev = torch.device("cuda", 0)
for m, s, n_seq in ((50, 50, 1), (50, 50, 4096), (100, 100, 1), (100, 100, 256), (100, 100, 4096), (300, 200, 4096), (1024, 100, 1024)):
    rng = np.random.RandomState(m + s)
    mm = np.cumsum(rng.uniform(-8, 10, m)) + 40; ms = rng.uniform(0.5, 3, m); md = rng.uniform(0.0005, 0.01, m)
    idx = np.clip(np.cumsum(rng.choice([0, 1, 1, 1, 2, -1], size=(n_seq, s)), axis=1) + 2, 1, m - 1)
    sm = mm[idx] + rng.normal(0, 0.2, (n_seq, s)); ss = rng.uniform(0.5, 3, (n_seq, s)); sd = rng.uniform(0.0005, 0.01, (n_seq, s))
    off = np.arange(n_seq + 1, dtype=np.int64) * s
    d = [torch.from_numpy(a.ravel()).to(dev) for a in (sm, ss, sd)]
    for _ in range(2):
        out = ctx.align_batch(mm, ms, md, 0.5, 0.5, d[0], d[1], d[2], off)
    torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 5
    for _ in range(reps):
        out = ctx.align_batch(mm, ms, md, 0.5, 0.5, d[0], d[1], d[2], off)
    torch.cuda.synchronize(); gpu = (time.perf_counter() - t0) / reps
    k = min(n_seq, 64); t0 = time.perf_counter()
    ref = [oracle.align_raw(mm, ms, md, 0.5, 0.5, sm[q], ss[q], sd[q]) for q in range(k)]
    cpu = (time.perf_counter() - t0) / k
    st = out[2].cpu().numpy(); sc = out[0].cpu().numpy(); pa = out[1].cpu().numpy().view(np.uint32).reshape(n_seq, s)
    ok = all(st[q] == ref[q][0] and (ref[q][0] != 0 or (sc[q] == ref[q][1] and np.array_equal(pa[q], ref[q][2]))) for q in range(k))
    print("m=%4d s=%4d n_seq=%5d  gpu %.3f ms/batch = %.2f us/alignment = %.1f Mcells/s | oracle %.1f us/alignment (1 core) | ok=%s status0=%d"
          % (m, s, n_seq, gpu * 1e3, gpu / n_seq * 1e6, n_seq * m * s / gpu / 1e6, cpu * 1e6, ok, int((st == 0).sum())))
