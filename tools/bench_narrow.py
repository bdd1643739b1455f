#!/usr/bin/env python3
"""Narrow-row sampler shapes on one MI355X: general-m rsample for m = 2 .. 16 through the C ABI, kernel time from
events.  These shapes are instruction-bound (a 128-row tile of m = 2 is 2.5 KiB), so this is the table to watch when the
rejection phase or the per-row arithmetic changes.  VMF_HIP_LIB selects the library build under test.
    python tools/bench_narrow.py [--shapes 2:27,2:24,4:24,10:24,16:24]"""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="2:27,2:24,3:24,4:24,8:24,10:24,12:24,16:24,20:24,31:24")
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--s1", type=int, default=27, help="log2 rows of the S^1 projected-normal case (0: skip)")
    args = ap.parse_args()
    lib = _lib.lib()
    sp = _lib.stream_ptr(DEV)
    dt = getattr(torch, args.dtype)
    es = 4 if dt == torch.float32 else 8
    g = torch.Generator(device=DEV).manual_seed(0)
    print(f"library: {os.environ.get('VMF_HIP_LIB', 'default')}")
    print("| m | rows | fwd ms | algorithmic GB/s | G rows/s | mean proposals | bwd ms | bwd GB/s |")
    print("|---|---|---|---|---|---|---|---|")
    for item in args.shapes.split(","):
        m, lg = (int(v) for v in item.split(":"))
        n = 1 << lg
        mu = torch.nn.functional.normalize(torch.randn(n, m, device=DEV, generator=g), dim=-1).to(dt)
        kap = torch.exp(torch.rand(n, device=DEV, generator=g) * np.log(1e4)).to(dt)
        z = torch.empty_like(mu)

        def f():
            rc = lib.vmf_rsample_fwd(_lib.ptr(mu), m, _lib.ptr(kap), 1, _lib.ptr(z), None, None, None, None, None, n, m,
                                     None, None, 0, None, 64, 0, 1, 0, None, 0, _lib.dtype_code(mu), sp)
            assert rc == 0, lib.vmf_last_error_string()
        ms = timeit(f)
        # mean number of proposals per row on a sample of the rows
        ns = min(n, 1 << 20)
        acc = torch.empty(ns, device=DEV, dtype=torch.int32)
        rc = lib.vmf_rsample_fwd(_lib.ptr(mu), m, _lib.ptr(kap), 1, _lib.ptr(z), None, None, _lib.ptr(acc), None, None, ns, m,
                                 None, None, 0, None, 64, 0, 1, 0, None, 0, _lib.dtype_code(mu), sp)
        assert rc == 0
        torch.cuda.synchronize()
        mean_prop = float(acc.float().mean()) + 1.0
        # backward: reads mu, grad_z, kappa, w, 1 - w; writes dmu, dkappa (noise regenerated)
        w, omw = torch.empty(n, device=DEV, dtype=dt), torch.empty(n, device=DEV, dtype=dt)
        rc = lib.vmf_rsample_fwd(_lib.ptr(mu), m, _lib.ptr(kap), 1, _lib.ptr(z), _lib.ptr(w), _lib.ptr(omw), None, None, None, n, m,
                                 None, None, 0, None, 64, 0, 1, 0, None, 0, _lib.dtype_code(mu), sp)
        assert rc == 0
        gz = torch.randn_like(mu)
        dmu, dk = torch.empty_like(mu), torch.empty_like(kap)

        def fb():
            rc = lib.vmf_rsample_bwd(_lib.ptr(mu), m, _lib.ptr(kap), 1, _lib.ptr(w), _lib.ptr(omw), _lib.ptr(gz), None,
                                     None, _lib.ptr(dmu), _lib.ptr(dk), n, m, 0, 1, 0, None, 0, _lib.dtype_code(mu), sp)
            assert rc == 0, lib.vmf_last_error_string()
        msb = timeit(fb)
        print(f"| {m} | 2^{lg} | {ms:.3f} | {n * (2 * m + 1) * es / ms / 1e6:.0f} | {n / ms / 1e6:.2f} | {mean_prop:.3f} "
              f"| {msb:.3f} | {n * (3 * m + 4) * es / msb / 1e6:.0f} |")
        del mu, z, kap, gz, dmu, dk, w, omw
        torch.cuda.empty_cache()
    if args.s1:
        n = 1 << args.s1
        mu = torch.nn.functional.normalize(torch.randn(n, 2, device=DEV, generator=g), dim=-1).to(dt)
        kap = (torch.rand(n, device=DEV, generator=g) * 50 + 1).to(dt)
        z = torch.empty_like(mu)
        gz = torch.randn_like(mu)
        dmu = torch.empty_like(mu)
        dk = torch.empty_like(kap)

        def f1():
            rc = lib.vmf_s1_projnormal_fwd(_lib.ptr(mu), _lib.ptr(kap), None, _lib.ptr(z), n, 2, 0, 1, 0, None,
                                           _lib.dtype_code(mu), sp)
            assert rc == 0

        def f2():
            rc = lib.vmf_s1_projnormal_bwd(_lib.ptr(mu), _lib.ptr(kap), None, _lib.ptr(gz), _lib.ptr(dmu), _lib.ptr(dk),
                                           n, 2, 0, 1, 0, None, _lib.dtype_code(mu), sp)
            assert rc == 0
        for name, f, b in (("S^1 projected normal fwd", f1, 5 * es), ("S^1 projected normal bwd", f2, 8 * es)):
            ms = timeit(f)
            print(f"| {name} | 2^{args.s1} | {ms:.3f} | {n * b / ms / 1e6:.0f} | {n / ms / 1e6:.2f} | - | - | - |")


if __name__ == "__main__":
    main()
