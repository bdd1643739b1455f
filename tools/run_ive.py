#!/usr/bin/env python3
"""a few launches of ive forward / backward for given orders, for rocprofv3 (--kernel-trace --stats or --pmc):
    python tools/run_ive.py 2 31 255 [--reps 3] [--hi 1e3] [--f32]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from medicalimageanalysis_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("orders", nargs="+", type=float)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--hi", type=float, default=1e3)
ap.add_argument("--f32", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
n = 1 << 24
g = torch.Generator(device=dev).manual_seed(0)
x = torch.exp(torch.rand(n, device=dev, generator=g, dtype=torch.float64) * np.log(a.hi))
if a.f32:
    x = x.float()
out, go = torch.empty_like(x), torch.ones_like(x)
lib, sp = _lib.lib(), _lib.stream_ptr(dev)
for v in a.orders:
    for _ in range(a.reps):
        assert lib.vmf_ive_fwd(v, _lib.ptr(x), _lib.ptr(out), n, _lib.dtype_code(x), sp) == 0
    for _ in range(a.reps):
        assert lib.vmf_ive_bwd(v, _lib.ptr(x), _lib.ptr(go), _lib.ptr(out), n, _lib.dtype_code(x), sp) == 0
torch.cuda.synchronize()
print("done")
