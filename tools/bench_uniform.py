#!/usr/bin/env python3
"""The uniform prior on S^dim (HypersphericalUniform.sample, in-kernel noise, fp32) on one MI355X: a write-only stream
of (dim + 1) floats per row.  Times include the Python call and the output allocation.
    python tools/bench_uniform.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VMF_RNG"] = "philox"

import torch  # noqa: E402

from _timing import timeit  # noqa: E402
from medicalimageanalysis_amd.sampler.spherical_uniform import HypersphericalUniform  # noqa: E402

DEV = torch.device("cuda:0")
print("| prior | rows | ms | GB/s written | of 8 TB/s |")
print("|---|---|---|---|---|")
for dim, lg in ((1, 26), (2, 24), (3, 24), (9, 22), (15, 22), (31, 21), (63, 20), (64, 20), (255, 18), (511, 17)):
    n = 1 << lg
    d = HypersphericalUniform(dim, device=DEV)
    ms = timeit(lambda: d.sample((n,)))
    gbs = n * (dim + 1) * 4 / ms / 1e6
    print(f"| S^{dim} | 2^{lg} | {ms:.4f} | {gbs:.0f} | {gbs / 80:.1f} % |", flush=True)
