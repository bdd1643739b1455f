#!/usr/bin/env python3
"""The path's secondary entry points on one MI355X, through the Python API (call overhead and output allocation
included): `_sample_w3` alone, `mean`, the CNN experiment's `log_prob`, general-m `log_prob` / `entropy` / KL against
the uniform prior, `log_ive`, the Bessel ratio, and a sample shape in front of shared parameters.
    python tools/bench_secondary.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VMF_RNG"] = "philox"

import numpy as np  # noqa: E402
import torch  # noqa: E402

from _timing import timeit  # noqa: E402
from medicalimageanalysis_amd import _lib  # noqa: E402
from medicalimageanalysis_amd.sampler.ive import bessel_ratio, ive, log_ive  # noqa: E402
from medicalimageanalysis_amd.sampler.von_mises_fisher import (VonMisesFisher, VonMisesFisher3D,  # noqa: E402
                                                                VonMisesFisher3DLogProb)

DEV = torch.device("cuda:0")
g = torch.Generator(device=DEV).manual_seed(0)
torch.distributions.Distribution.set_default_validate_args(False)


def row(name, ms, nbytes):
    print(f"| {name} | {ms:.4f} | {nbytes / ms / 1e6:.0f} | {nbytes / ms / 8e7:.1f} % |", flush=True)


print("| case | ms | algorithmic GB/s | of 8 TB/s |")
print("|---|---|---|---|")
n = 1 << 24
loc = torch.nn.functional.normalize(torch.randn(n, 3, device=DEV, generator=g), dim=-1)
kap = torch.rand(n, device=DEV, generator=g) * 99 + 1
q = VonMisesFisher3D(loc, kap)
row("VonMisesFisher3D._sample_w3, 2^24 fp32 (4 B in, 4 B out)", timeit(lambda: q._sample_w3(())), n * 8)
row("VonMisesFisher3D.mean_fused, 2^24 fp32 (16 B in, 12 B out)", timeit(lambda: q.mean_fused()), n * 28)
row("VonMisesFisher3D.mean() (reference formula: two ive calls + torch ops), 2^24 fp32", timeit(lambda: q.mean()), n * 28)
ql = VonMisesFisher3DLogProb(loc, kap)
x = torch.nn.functional.normalize(torch.randn(n, 3, device=DEV, generator=g), dim=-1)
row("VonMisesFisher3DLogProb.log_prob, 2^24 fp32 (28 B in, 4 B out)", timeit(lambda: ql.log_prob(x)), n * 32)
del q, ql, x, loc
for m, lg in ((10, 24), (64, 22)):
    nn = 1 << lg
    mu = torch.nn.functional.normalize(torch.randn(nn, m, device=DEV, generator=g), dim=-1)
    kk = torch.exp(torch.rand(nn, device=DEV, generator=g) * np.log(1e4))
    d = VonMisesFisher(mu, kk)
    z = torch.nn.functional.normalize(torch.randn(nn, m, device=DEV, generator=g), dim=-1)
    row(f"VonMisesFisher(m={m}).log_prob, 2^{lg} fp32", timeit(lambda: d.log_prob(z)), nn * (2 * m + 2) * 4)
    gl, dmu, dx, dk = torch.randn(nn, device=DEV, generator=g), torch.empty_like(mu), torch.empty_like(mu), torch.empty_like(kk)
    sp = _lib.stream_ptr(DEV)
    row(f"vmf_log_prob_bwd (d mu, d x, d kappa), m={m}, 2^{lg} fp32",
        timeit(lambda: _lib.lib().vmf_log_prob_bwd(_lib.ptr(mu), m, _lib.ptr(kk), 1, _lib.ptr(z), _lib.ptr(gl), _lib.ptr(dmu),
                                                   _lib.ptr(dx), _lib.ptr(dk), nn, m, 0, sp)), nn * (4 * m + 3) * 4)
    del gl, dmu, dx, dk
    row(f"VonMisesFisher(m={m}).entropy, 2^{lg} fp32 (4 B in, 4 B out)", timeit(lambda: d.entropy()), nn * 8)
    row(f"VonMisesFisher(m={m}).kl_uniform, 2^{lg} fp32", timeit(lambda: d.kl_uniform()), nn * 8)
    del mu, kk, d, z
k64 = torch.exp(torch.rand(n, device=DEV, generator=g, dtype=torch.float64) * np.log(1e4))
for v in (4.0, 31.0, 255.0):
    row(f"log_ive({v:g}, z), 2^24 fp64", timeit(lambda: log_ive(v, k64)), n * 16)
    row(f"bessel_ratio({v:g}, z), 2^24 fp64", timeit(lambda: bessel_ratio(v, k64)), n * 16)
k32 = k64.float()
row("bessel_ratio(5, z), 2^24 fp32", timeit(lambda: bessel_ratio(5.0, k32)), n * 8)
row("ive(1.5, z), 2^24 fp32", timeit(lambda: ive(1.5, k32)), n * 8)
# a sample shape in front of shared parameters (the GENERAL broadcast kind of the m = 3 entry points)
loc1 = torch.nn.functional.normalize(torch.randn(1, 3, device=DEV, generator=g), dim=-1)
q1 = VonMisesFisher3D(loc1, torch.tensor([10.0], device=DEV))
ns = 1 << 22
row("VonMisesFisher3D(loc (1,3), scale (1,)).rsample((2^22,)) fp32 (12 B out)", timeit(lambda: q1.rsample((ns,))), ns * 12)
