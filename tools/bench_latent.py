#!/usr/bin/env python3
"""Latent-loss and torus-embedding kernels alone on one MI355X: forward (mean only / with per-row output) and backward,
2^24 rows of 3 against 4096 label rows, fp32.  Prints a markdown table; run on the GPU box:  python tools/bench_latent.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from _timing import timeit as _timeit  # noqa: E402

from medicalimageanalysis_amd import _lib  # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, reps=30, warm=None):
    return _timeit(fn, reps=reps)


def main():
    lib = _lib.lib()
    sp = _lib.stream_ptr(DEV)
    g = torch.Generator(device=DEV).manual_seed(0)
    n, B = 1 << 24, 4096
    z = torch.randn(n, 3, device=DEV, generator=g)
    zt = z * 0.3 + torch.tensor([1.2, 0.9, 0.0], device=DEV)
    labels = torch.rand(B, 2, device=DEV, generator=g) * 3.0
    rows = torch.empty(n, device=DEV)
    mean = torch.empty((), device=DEV)
    gm = torch.ones((), device=DEV)
    dz = torch.empty_like(z)
    ws = torch.empty(_lib.KL_WORKSPACE_DOUBLES, device=DEV, dtype=torch.float64)
    print("| case (2^24 rows, fp32) | ms | algorithmic GB/s | % of 8 TB/s |")
    print("|---|---|---|---|")
    for v, name in enumerate(("S1", "S2", "T2")):
        zz = zt if name == "T2" else z

        def fwd(per_row):
            rc = lib.vmf_latent_loss_fwd(_lib.ptr(zz), _lib.ptr(labels), _lib.ptr(rows) if per_row else None, _lib.ptr(mean),
                                         _lib.ptr(ws), n, 3, B, 2, v, 1.5, 0.5, _lib.VMF_F32, sp)
            assert rc == 0

        def bwd():
            rc = lib.vmf_latent_loss_bwd(_lib.ptr(zz), _lib.ptr(labels), _lib.ptr(gm), _lib.ptr(dz), n, 3, B, 2, v, 1.5,
                                         0.5, _lib.VMF_F32, sp)
            assert rc == 0
        for label, fn, nb in ((f"latent_loss {name} fwd, mean only", lambda: fwd(False), 12),
                              (f"latent_loss {name} fwd + per-row output", lambda: fwd(True), 16),
                              (f"latent_loss {name} bwd", bwd, 24)):
            ms = timeit(fn)
            gbs = n * nb / ms / 1e6
            print(f"| {label} | {ms:.4f} | {gbs:.0f} | {gbs / 80:.1f} |")


if __name__ == "__main__":
    main()
