#!/usr/bin/env python3
"""Randomised sweep of ive / its derivative / log_ive / the Bessel ratio on the GPU against scipy.special.ive -- the
function the reference itself calls (sampler/ive.py:14-33: forward scipy's ive on the host, backward
ive(v-1, z) - ive(v, z) (v + z) / z).  Orders: the ones the hot path uses (m/2 - 1 and m/2 for m = 2 .. 2048), random
reals in [0, 400] and orders next to the regime switches; arguments log-uniform over [1e-6, 1e6] plus values next to
the switches.  Test infrastructure.  python tests/fuzz_ive.py [--orders 150] [--seed 0]"""
import argparse
import os
import sys

import numpy as np
import scipy.special as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from medicalimageanalysis_amd.sampler.ive import bessel_ratio, ive, log_ive  # noqa: E402

DEV = torch.device("cuda:0")


def sweep(n_orders, seed0, n_x=20000):
    gen = np.random.default_rng(seed0)
    worst = {"ive": (0.0, None), "dive": (0.0, None), "log_ive": (0.0, None), "ratio": (0.0, None), "ive32": (0.0, None)}
    fixed = [0.0, 0.5, 1.0, 1.5, 2.0, 4.0, 4.5, 5.0, 31.0, 32.0, 255.0, 256.0, 383.0, 384.0, 1023.0, 1024.0]
    orders = fixed + list(gen.uniform(0, 400, n_orders)) + list(gen.integers(2, 300, n_orders // 3).astype(float))
    for v in orders:
        v = float(v)
        x = np.exp(gen.uniform(np.log(1e-6), np.log(1e6), n_x))
        # arguments around the order (where series, continued fraction and uniform expansion hand over)
        x[: n_x // 4] = (v + 1.0) * np.exp(gen.uniform(np.log(0.05), np.log(40.0), n_x // 4))
        ref = sp.ive(v, x)
        ok = np.isfinite(ref) & (ref > 1e-300)  # below: underflow territory, compared through log_ive
        tx = torch.from_numpy(x).to(DEV).requires_grad_()
        y = ive(v, tx)
        y.sum().backward()
        got = y.detach().cpu().numpy()
        e = np.abs(got[ok] - ref[ok]) / ref[ok]
        if e.size and e.max() > worst["ive"][0]:
            worst["ive"] = (float(e.max()), (v, float(x[ok][e.argmax()])))
        # the reference's own backward formula, from scipy in double
        dref = sp.ive(v - 1, x) - ref * (v + x) / x if v >= 1 else None
        if dref is not None:
            okd = ok & np.isfinite(dref)
            # the formula is a difference of nearly equal terms at large x: compare relative to the terms
            scale = np.abs(sp.ive(v - 1, x)) + np.abs(ref * (v + x) / x)
            ed = np.abs(tx.grad.cpu().numpy()[okd] - dref[okd]) / scale[okd]
            if ed.size and ed.max() > worst["dive"][0]:
                worst["dive"] = (float(ed.max()), (v, float(x[okd][ed.argmax()])))
        lg = log_ive(v, tx.detach()).cpu().numpy()
        el = np.abs(lg[ok] - np.log(ref[ok])) / (1.0 + np.abs(np.log(ref[ok])))
        if el.size and el.max() > worst["log_ive"][0]:
            worst["log_ive"] = (float(el.max()), (v, float(x[ok][el.argmax()])))
        if v >= 1:  # I_v / I_{v-1}
            den = sp.ive(v - 1, x)
            okr = ok & np.isfinite(den) & (den > 1e-300)
            rr = bessel_ratio(v, tx.detach()).cpu().numpy()
            rref = ref[okr] / den[okr]
            er = np.abs(rr[okr] - rref) / rref
            if er.size and er.max() > worst["ratio"][0]:
                worst["ratio"] = (float(er.max()), (v, float(x[okr][er.argmax()])))
        # fp32 tensors: double arithmetic inside, one rounding at the end (scipy's ff->f loop does the same)
        x32 = x.astype(np.float32)
        v32 = float(np.float32(v))  # scipy's float loop rounds the order too; the hot path's orders are half-integers
        ref32 = sp.ive(np.float32(v), x32).astype(np.float64)
        ok32 = np.isfinite(ref32) & (ref32 > 1e-30)
        got32 = ive(v32, torch.from_numpy(x32).to(DEV)).cpu().numpy().astype(np.float64)
        e32 = np.abs(got32[ok32] - ref32[ok32]) / ref32[ok32]
        if e32.size and e32.max() > worst["ive32"][0]:
            worst["ive32"] = (float(e32.max()), (v, float(x32[ok32][e32.argmax()])))
    return {"orders": len(orders), "points_per_order": n_x, "worst": worst}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--orders", type=int, default=150)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    print(sweep(a.orders, a.seed))
