#!/usr/bin/env python3
"""Host-side cost per call of the Python entry points at the reference's batch sizes (the GPU work is microseconds
there, so the wall time of an eager train step is this).  Calls are issued back to back without synchronising; the
time per call is what the host needs to validate, plan and launch.  python tools/host_overhead.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medicalimageanalysis_amd import kl as klmod, losses  # noqa: E402
from medicalimageanalysis_amd.sampler.ive import ive  # noqa: E402
from medicalimageanalysis_amd.sampler.von_mises_fisher import (VonMisesFisher, VonMisesFisher3D,  # noqa: E402
                                                                vmf3_rsample_from_heads)

DEV = torch.device("cuda:0")
B = 128
loc = torch.randn(B, 3, device=DEV, requires_grad=True)
raw = torch.randn(B, 1, device=DEV, requires_grad=True)
kap = torch.nn.functional.softplus(raw.detach()) + 1
kap1 = kap.squeeze(-1).clone()
mu10 = torch.nn.functional.normalize(torch.randn(B, 10, device=DEV), dim=-1)
x = torch.rand(B, 784, device=DEV)
xh = torch.rand(B, B, 784, device=DEV, requires_grad=True)


def per_call(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e6


def fwd_bwd():
    z, k = vmf3_rsample_from_heads(loc, raw)
    (z.sum() + k.sum()).backward()


print("| call (batch 128) | host microseconds per call |")
print("|---|---|")
for mode in ("torch", "philox"):
    os.environ["VMF_RNG"] = mode
    print(f"| VonMisesFisher3D(loc, (B,1) kappa).rsample(), VMF_RNG={mode} | "
          f"{per_call(lambda: VonMisesFisher3D(loc.detach(), kap, validate_args=False).rsample()):.1f} |")
    print(f"| VonMisesFisher3D(loc, (B,) kappa).rsample(), VMF_RNG={mode} | "
          f"{per_call(lambda: VonMisesFisher3D(loc.detach(), kap1, validate_args=False).rsample()):.1f} |")
    print(f"| vmf3_rsample_from_heads forward + backward, VMF_RNG={mode} | {per_call(fwd_bwd, 500):.1f} |")
print(f"| VonMisesFisher(mu (B,10), kappa).rsample() | "
      f"{per_call(lambda: VonMisesFisher(mu10, kap1, validate_args=False).rsample()):.1f} |")
print(f"| kl_mean(kappa, spherical) | {per_call(lambda: klmod.kl_mean(kap, 'spherical')):.1f} |")
print(f"| ive(1.5, kappa) | {per_call(lambda: ive(1.5, kap)):.1f} |")
print(f"| elbo_loss(x, x_hat (B,B,784), kappa) forward | "
      f"{per_call(lambda: losses.elbo_loss(x, xh.detach(), kap, 1.0, 1.0, 'spherical')):.1f} |")
print(f"| torch.rand + torch.randn of the same shapes (for scale) | "
      f"{per_call(lambda: (torch.rand(B, 1, device=DEV), torch.randn(B, 3, device=DEV))):.1f} |")
