#!/usr/bin/env python3
"""The fused KL reductions (2^27 fp32 elements, 4 B of HBM traffic each) through the C ABI -- kernel time from HIP events
around the two launches (block partials + finish), nothing allocated inside the timed call -- and ive for a few orders.
    python tools/bench_kl.py [--only spherical,s1,stable_m2] [--reps 10] [--no-ive]
Under rocprofv3 (--kernel-trace --stats or --pmc) use --reps 3 --no-ive."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench_extra import timeit  # noqa: E402
from medicalimageanalysis_amd import _lib, kl as klmod  # noqa: E402
from medicalimageanalysis_amd.sampler.ive import ive  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--only", default="spherical,torus,circular,s1,stable_m3,stable_m2")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--no-ive", action="store_true")
args = ap.parse_args()

DEV = torch.device("cuda:0")
g = torch.Generator(device=DEV).manual_seed(0)
n = 1 << 27
kap = torch.exp(torch.rand(n, device=DEV, generator=g) * np.log(1e3))
# the reference's unguarded forms overflow to NaN beyond kappa ~ 89 (sinh) / ~ 92 (I0): they are timed on the range they
# are defined on
kap80 = torch.exp(torch.rand(n, device=DEV, generator=g) * np.log(80.0))
lib, sp = _lib.lib(), _lib.stream_ptr(DEV)
ws = torch.empty(_lib.KL_WORKSPACE_DOUBLES, device=DEV, dtype=torch.float64)
acc = torch.empty(1, device=DEV, dtype=torch.float64)
print("| case | ms | GB/s | of 8 TB/s |")
print("|---|---|---|---|")
for v in args.only.split(","):
    data, rng_ = (kap80, "[1,80]") if v in ("torus", "circular", "s1") else (kap, "[1,1e3]")

    def f():
        rc = lib.vmf_kl_fwd(_lib.ptr(data), None, _lib.ptr(acc), _lib.ptr(ws), 1.0, n, klmod.VARIANTS[v], _lib.VMF_F32, sp)
        assert rc == 0
    ms = timeit(f, reps=args.reps, warm=(args.reps if args.reps < 10 else None))
    print(f"| KL sum {v} fp32 2^27, kappa log-uniform {rng_} | {ms:.3f} | {n * 4 / ms / 1e6:.0f} | {n * 4 / ms / 8e7:.1f} % |",
          flush=True)
# fused ELBO through the C ABI: x (B, D) against x_hat (B, B, D), the reference's broadcast (model/spherical_vae.py:133-138)
for B in (128, 1024):
    D = 784
    x = torch.rand(B, D, device=DEV, generator=g)
    xh = torch.rand(B, B, D, device=DEV, generator=g)
    kp = torch.rand(B, 1, device=DEV, generator=g) * 50 + 1
    loss = torch.empty((), device=DEV)
    wse = torch.empty(_lib.ELBO_WORKSPACE_DOUBLES, device=DEV, dtype=torch.float64)
    dxh, dkp, one1 = torch.empty_like(xh), torch.empty_like(kp), torch.ones((), device=DEV)

    def ef():
        rc = lib.vmf_elbo_fwd(_lib.ptr(x), _lib.ptr(xh), _lib.ptr(kp), _lib.ptr(loss), None, _lib.ptr(wse), 1.0 / xh.numel(),
                              1.0 / B, x.numel(), xh.numel(), B, klmod.VARIANTS["spherical"], _lib.VMF_F32, sp)
        assert rc == 0

    def eb():
        rc = lib.vmf_elbo_bwd(_lib.ptr(x), _lib.ptr(xh), _lib.ptr(kp), _lib.ptr(one1), _lib.ptr(dxh), _lib.ptr(dkp),
                              1.0 / xh.numel(), 1.0 / B, x.numel(), xh.numel(), B, klmod.VARIANTS["spherical"], _lib.VMF_F32, sp)
        assert rc == 0
    ms = timeit(ef, reps=args.reps, warm=(args.reps if args.reps < 10 else None))
    print(f"| fused ELBO forward, x_hat ({B},{B},{D}) fp32 | {ms:.4f} | {xh.numel() * 4 / ms / 1e6:.0f} | "
          f"{xh.numel() * 4 / ms / 8e7:.1f} % |", flush=True)
    ms = timeit(eb, reps=args.reps, warm=(args.reps if args.reps < 10 else None))
    print(f"| fused ELBO backward, x_hat ({B},{B},{D}) fp32 | {ms:.4f} | {xh.numel() * 8 / ms / 1e6:.0f} | "
          f"{xh.numel() * 8 / ms / 8e7:.1f} % |", flush=True)
    del x, xh, dxh
if not args.no_ive:
    kg = kap[: 1 << 24].clone()
    dk = torch.empty_like(kg)
    one = torch.ones(1, device=DEV)

    def bw():
        rc = lib.vmf_kl_bwd(_lib.ptr(kg), _lib.ptr(one), 0, 1.0 / kg.numel(), _lib.ptr(dk), kg.numel(),
                            klmod.VARIANTS["spherical"], _lib.VMF_F32, sp)
        assert rc == 0
    ms = timeit(bw, reps=10)
    print(f"| KL mean backward spherical fp32 2^24 | {ms:.3f} | {(1 << 24) * 8 / ms / 1e6:.0f} | |")
    x = kap[: 1 << 24].double()
    for v in (0, 1, 0.5, 2.0):
        ms = timeit(lambda: ive(v, x), reps=10)
        print(f"| ive({v}) fp64 2^24 | {ms:.3f} | {(1 << 24) * 16 / ms / 1e6:.0f} | |")
