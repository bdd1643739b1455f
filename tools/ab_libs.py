#!/usr/bin/env python3
"""A/B of several builds of libvmf_hip.so on the headline workload in ONE process, interleaved round-robin so that
clock drift and box-to-box differences cancel:  python tools/ab_libs.py name=path [name=path ...] [--m 64 --rows N]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medicalimageanalysis_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--m", type=int, default=64)
ap.add_argument("--rows", type=int, default=1 << 24)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--no-log-prob", action="store_true")
ap.add_argument("--kernel", default="fwd", choices=["fwd", "bwd", "logprob", "logprobbwd", "s1fwd", "s1bwd"])
a = ap.parse_args()
dev = torch.device("cuda:0")
n, m = a.rows, a.m
mu = torch.nn.functional.normalize(torch.randn(n, m, device=dev), dim=-1)
kappa = torch.exp(torch.rand(n, device=dev) * float(np.log(1e4)))
z = torch.empty_like(mu)
logp = None if a.no_log_prob else torch.empty(n, device=dev)
st = torch.cuda.current_stream(dev)
sp = C.c_void_p(st.cuda_stream)
hs = []
for spec in a.libs:
    name, path = spec.split("=", 1)
    h = C.CDLL(os.path.abspath(path))
    h.vmf_rsample_fwd.argtypes = _lib._PROTOS["vmf_rsample_fwd"]
    h.vmf_rsample_fwd.restype = C.c_int
    for fn in ("vmf_rsample_bwd", "vmf_log_prob_fwd", "vmf_log_prob_bwd", "vmf_s1_projnormal_fwd", "vmf_s1_projnormal_bwd"):
        getattr(h, fn).argtypes = _lib._PROTOS[fn]
        getattr(h, fn).restype = C.c_int
    hs.append((name, h))


w = torch.empty(n, device=dev)
omw = torch.empty(n, device=dev)
dk = torch.empty(n, device=dev)
if a.kernel.startswith("s1"):
    gz = torch.randn(n, m, device=dev)
    dmu = torch.empty_like(mu)
elif a.kernel != "fwd":
    rc = hs[0][1].vmf_rsample_fwd(_lib.ptr(mu), m, _lib.ptr(kappa), 1, _lib.ptr(z), _lib.ptr(w), _lib.ptr(omw), None, None,
                                  None, n, m, None, None, 0, None, 64, 0, 0x5EED, 0, None, 0, _lib.VMF_F32, sp)
    assert rc == 0
    gz = torch.randn(n, m, device=dev)
    dmu = torch.empty_like(mu)
    logp = torch.randn(n, device=dev)


def step(h, i):
    if a.kernel == "s1fwd":
        rc = h.vmf_s1_projnormal_fwd(_lib.ptr(mu), _lib.ptr(kappa), None, _lib.ptr(z), n, m, 0, 0x5EED, i, None,
                                     _lib.VMF_F32, sp)
        assert rc == 0, rc
        return
    if a.kernel == "s1bwd":
        rc = h.vmf_s1_projnormal_bwd(_lib.ptr(mu), _lib.ptr(kappa), None, _lib.ptr(gz), _lib.ptr(dmu), _lib.ptr(dk), n, m, 0,
                                     0x5EED, i, None, _lib.VMF_F32, sp)
        assert rc == 0, rc
        return
    if a.kernel == "bwd":
        rc = h.vmf_rsample_bwd(_lib.ptr(mu), m, _lib.ptr(kappa), 1, _lib.ptr(w), _lib.ptr(omw), _lib.ptr(gz), None,
                               None, _lib.ptr(dmu), _lib.ptr(dk), n, m, 0, 0x5EED, 0, None, 0, _lib.VMF_F32, sp)
        assert rc == 0, rc
        return
    if a.kernel == "logprobbwd":
        rc = h.vmf_log_prob_bwd(_lib.ptr(mu), m, _lib.ptr(kappa), 1, _lib.ptr(z), _lib.ptr(logp), _lib.ptr(dmu), _lib.ptr(gz),
                                _lib.ptr(dk), n, m, _lib.VMF_F32, sp)
        assert rc == 0, rc
        return
    if a.kernel == "logprob":
        rc = h.vmf_log_prob_fwd(_lib.ptr(mu), m, _lib.ptr(kappa), 1, _lib.ptr(z), _lib.ptr(logp), n, m, _lib.VMF_F32, sp)
        assert rc == 0, rc
        return
    rc = h.vmf_rsample_fwd(_lib.ptr(mu), m, _lib.ptr(kappa), 1, _lib.ptr(z), None, None, None, _lib.ptr(logp), None, n, m, None,
                           None, 0, None, 64, 0, 0x5EED, i, None, 0, _lib.VMF_F32, sp)
    assert rc == 0, rc


for i in range(20):
    step(hs[0][1], i)
res = {name: [] for name, _ in hs}
for r in range(a.rounds):
    for name, h in hs:
        for i in range(3):
            step(h, i)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
        for i, (s, e) in enumerate(evs):
            s.record(st)
            step(h, 100 + i)
            e.record(st)
        torch.cuda.synchronize()
        res[name].append(float(np.mean([s.elapsed_time(e) for s, e in evs])))
for name, v in res.items():
    # the GPU takes tens of milliseconds of load to settle (the first rounds run 5-15 % slower): the last half counts
    print(f"{name:8s} settled {np.mean(v[len(v) // 2:]):.4f} ms  mean {np.mean(v):.4f}  rounds "
          + " ".join(f"{x:.4f}" for x in v), flush=True)
