"""Randomised check of ps_filter_bessel (GPU box): fused-halo kernel and exact three-pass scan against the oracle
(C restatement of scipy's bessel(1) + filtfilt) over random lengths, cutoffs, sampling rates, fp32 / int16 input."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
from pypore_amd import engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
ctx = engine.context(0)
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0; worst = 0.0; t0 = time.time(); fused_cases = 0
for seed in range(n_seeds):
    rng = np.random.RandomState(70_000 + seed)
    n = int(rng.choice([7, 8, 13, 100, 3000, 4096 - 12, 4097, int(rng.randint(7, 200_000)), int(rng.randint(7, 2_000_000))]))
    fs = float(rng.choice([1e4, 5e4, 1e5, 2.5e5]))
    cutoff = float(np.exp(rng.uniform(np.log(fs * 2e-5), np.log(fs * 0.24))))
    k = synth.random_dwell_counts(n, seed, 2, 4) if n < 100 else synth.random_dwell_counts(n, seed, 50, 5000)
    use_i16 = bool(rng.randint(0, 2))
    dev = torch.from_numpy(k.astype(np.int16)).cuda() if use_i16 else torch.from_numpy((k * synth.QUANTUM).astype(np.float32)).cuda()
    ref = oracle.bessel_filtfilt(k * synth.QUANTUM, cutoff, fs)
    scale = max(np.max(np.abs(ref)), 1e-300)
    for mode in (1, 0):
        ctx.set_option("filter_fused", mode)
        got = ctx.filter_bessel(dev, synth.QUANTUM, cutoff=cutoff, sampling_freq=fs).cpu().numpy()
        err = float(np.max(np.abs(got - ref)) / scale)
        worst = max(worst, err)
        if not err <= 1e-10:
            bad += 1
            print("MISMATCH seed %d n %d fs %g cutoff %g i16 %s fused %d: rel err %.3e" % (seed, n, fs, cutoff, use_i16, mode, err))
ctx.set_option("filter_fused", 1)
print("filter fuzz: %d seeds x 2 paths, %d problems, worst relative error %.2e, %.0f s" % (n_seeds, bad, worst, time.time() - t0))
