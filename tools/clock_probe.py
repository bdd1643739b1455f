#!/usr/bin/env python3
"""How the headline launch time settles after an idle period: per-launch times of 30 launches from idle, with nothing,
a few, or many plain device copies of the same buffers in front.  python tools/clock_probe.py"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medicalimageanalysis_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
n, m = 1 << 24, 64
mu = torch.nn.functional.normalize(torch.randn(n, m, device=dev), dim=-1)
kappa = torch.exp(torch.rand(n, device=dev) * float(np.log(1e4)))
z = torch.empty_like(mu)
logp = torch.empty(n, device=dev)
sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
lib = _lib.lib()


def launches(k, base):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
    for i, (s, e) in enumerate(ev):
        s.record()
        rc = lib.vmf_rsample_fwd(_lib.ptr(mu), m, _lib.ptr(kappa), 1, _lib.ptr(z), None, None, None, _lib.ptr(logp), None, n, m,
                                 None, None, 0, None, 64, 0, 0x5EED, base + i, None, 0, _lib.VMF_F32, sp)
        assert rc == 0
        e.record()
    torch.cuda.synchronize()
    return [s.elapsed_time(e) for s, e in ev]


torch.cuda.synchronize()
for name, copies, idle in (("idle 2 s, no copies", 0, 2.0), ("idle 2 s, 6 copies", 6, 2.0), ("idle 2 s, 40 copies", 40, 2.0),
                           ("idle 2 s, 120 copies", 120, 2.0), ("idle 0.2 s, no copies", 0, 0.2),
                           ("idle 2 s, no copies", 0, 2.0)):
    time.sleep(idle)
    for _ in range(copies):
        z.copy_(mu)
    t = launches(30, 1000)
    print(f"{name:24s} first 5+: " + " ".join(f"{x:.2f}" for x in t[:14]) + f" | mean 6..25 {np.mean(t[5:25]):.4f} "
          f"mean 21..30 {np.mean(t[20:]):.4f}", flush=True)
