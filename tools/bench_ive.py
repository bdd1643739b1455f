#!/usr/bin/env python3
"""ive forward / backward through the C ABI on one MI355X: 2^24 elements, fp64 and fp32, a few orders.
    python tools/bench_ive.py        (prints a markdown table; kernel time from events around the launch)"""
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
    lib, sp = _lib.lib(), _lib.stream_ptr(DEV)
    n = 1 << 24
    g = torch.Generator(device=DEV).manual_seed(0)
    print("| case | ms | algorithmic GB/s | of 8 TB/s |")
    print("|---|---|---|---|")
    for dt in (torch.float64, torch.float32):
        es = 8 if dt == torch.float64 else 4
        for hi in (1e3, 1e5):
            x = torch.exp(torch.rand(n, device=DEV, generator=g, dtype=torch.float64) * np.log(hi)).to(dt)
            out, go = torch.empty_like(x), torch.ones_like(x)
            for v in (0.0, 1.0, 0.5, 2.0, 4.5, 31.0, 255.0, 255.5):
                def f():
                    assert lib.vmf_ive_fwd(v, _lib.ptr(x), _lib.ptr(out), n, _lib.dtype_code(x), sp) == 0
                ms = timeit(f)
                print(f"| ive({v:g}) fwd, 2^24 {str(dt)[6:]}, z log-uniform [1,{hi:g}] | {ms:.3f} | {n * 2 * es / ms / 1e6:.0f} | "
                      f"{n * 2 * es / ms / 8e7:.1f} % |", flush=True)
            for v in (2.0, 31.0, 255.0):
                def b():
                    assert lib.vmf_ive_bwd(v, _lib.ptr(x), _lib.ptr(go), _lib.ptr(out), n, _lib.dtype_code(x), sp) == 0
                ms = timeit(b)
                print(f"| ive({v:g}) bwd, 2^24 {str(dt)[6:]}, z log-uniform [1,{hi:g}] | {ms:.3f} | {n * 3 * es / ms / 1e6:.0f} | "
                      f"{n * 3 * es / ms / 8e7:.1f} % |", flush=True)


if __name__ == "__main__":
    main()
