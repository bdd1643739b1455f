#!/usr/bin/env python3
"""Time the fused KL reductions (2^27 fp32 elements) and ive for a few orders on one MI355X."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench_extra import timeit  # noqa: E402
from medicalimageanalysis_amd import kl as klmod  # noqa: E402
from medicalimageanalysis_amd.sampler.ive import ive  # noqa: E402

DEV = torch.device("cuda:0")
g = torch.Generator(device=DEV).manual_seed(0)
n = 1 << 27
kap = torch.exp(torch.rand(n, device=DEV, generator=g) * np.log(1e3))
print("| case | ms | GB/s |")
print("|---|---|---|")
for v in ("spherical", "torus", "circular", "s1", "stable_m3", "stable_m2"):
    ms = timeit(lambda: klmod.kl_sum(kap, v), reps=10, warm=10)
    print(f"| KL sum {v} fp32 2^27 | {ms:.3f} | {n * 4 / ms / 1e6:.0f} |")
kg = kap[: 1 << 24].clone().requires_grad_()
loss = klmod.kl_mean(kg, "spherical")


def bw():
    kg.grad = None
    loss.backward(retain_graph=True)


ms = timeit(bw, reps=10, warm=10)
print(f"| KL mean backward spherical fp32 2^24 | {ms:.3f} | {(1 << 24) * 8 / ms / 1e6:.0f} |")
x = kap[: 1 << 24].double()
for v in (0, 1, 0.5, 2.0):
    ms = timeit(lambda: ive(v, x), reps=10, warm=10)
    print(f"| ive({v}) fp64 2^24 | {ms:.3f} | {(1 << 24) * 16 / ms / 1e6:.0f} |")
