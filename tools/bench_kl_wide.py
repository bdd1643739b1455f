#!/usr/bin/env python3
"""The S^1 KL sum (reference form: NaN once e^kappa overflows) on concentrations up to 1e3, where a third of the batch
lies beyond the overflow point.  Run on the GPU box:  python tools/bench_kl_wide.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from medicalimageanalysis_amd import kl as klmod  # noqa: E402

DEV = torch.device("cuda:0")
n = 1 << 27
g = torch.Generator(device=DEV).manual_seed(0)
for hi in (80.0, 1e3):
    kap = torch.exp(torch.rand(n, device=DEV, generator=g) * np.log(hi))
    for variant in ("s1", "stable_m2", "spherical", "torus", "circular", "stable_m3"):
        for _ in range(10):
            klmod.kl_sum(kap, variant)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in ev:
            a.record()
            klmod.kl_sum(kap, variant)
            b.record()
        torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
        print(f"| KL sum {variant} fp32 2^27, kappa log-uniform [1,{hi:g}] | {ms:.3f} | {n * 4 / ms / 1e6:.0f} GB/s |")
