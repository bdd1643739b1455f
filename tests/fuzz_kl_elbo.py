#!/usr/bin/env python3
"""Randomised sweep of the KL and fused-ELBO kernels on the GPU against the CPU oracle (test infrastructure; the GPU
suite runs a short sweep through `sweep`): random element counts (odd, tiny, multi-block), misaligned views, all six
variants, both dtypes, concentrations over eleven decades with the special values the reference's formulas see
(0, negative, inf, NaN, values around every region switch); per-element values with NaN / inf positions exact, the
fused sums, the gradients, and the fused ELBO over random (rows, features) shapes with x broadcast against x_hat.
python tests/fuzz_kl_elbo.py [--cases 300] [--seed 0]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from medicalimageanalysis_amd.kl import VARIANTS, kl_mean, kl_per_sample, kl_sum  # noqa: E402
from medicalimageanalysis_amd.losses import elbo_loss  # noqa: E402
from oracle import consumers_oracle as co  # noqa: E402
from oracle import vmf_oracle as orc  # noqa: E402

DEV = torch.device("cuda:0")
SWITCHES = [1e-8, 1e-3, 1.0, 3.0, 20.0, 85.0, 88.7, 700.0, 1000.0]


def _view(gen, a):
    """the values of `a` on the device behind an aligned or a misaligned base"""
    t = torch.from_numpy(a).to(DEV)
    if gen.random() < 0.5 or t.numel() == 0:
        return t
    buf = torch.empty(t.numel() + 1, device=DEV, dtype=t.dtype)
    buf[1:] = t.reshape(-1)
    return buf[1:].view(t.shape)


def sweep(cases, seed0):
    gen = np.random.default_rng(seed0)
    worst = {"kl64": 0.0, "kl32": 0.0, "dkl64": 0.0, "dkl32": 0.0, "sum": 0.0, "elbo": 0.0, "delbo": 0.0}
    names = list(VARIANTS)
    for case in range(cases):
        name = names[int(gen.integers(0, len(names)))]
        dt = np.float64 if gen.random() < 0.5 else np.float32
        n = int(gen.choice([1, 2, 3, 5, 63, 64, 65, 255, 1000, 4097, 65537, 300001]))
        k = np.exp(gen.uniform(np.log(1e-6), np.log(1e5), n))
        nsw = min(n, 64)
        k[:nsw] = np.array(SWITCHES)[gen.integers(0, len(SWITCHES), nsw)] * (1 + gen.uniform(-1e-3, 1e-3, nsw))
        special = case % 3 == 0 and n >= 8
        if special:
            k[-6:] = [0.0, -1.0, np.inf, np.nan, 1e-30, 3e38 if dt == np.float32 else 1e300]
        k = k.astype(dt)
        tag = f"case {case}: {name} {dt.__name__} n={n} special={special}"
        with np.errstate(all="ignore"):
            ref, dref, sref = orc.vmf_kl(k, VARIANTS[name])
        tk = _view(gen, k).requires_grad_()
        out = kl_per_sample(tk, name)
        got = out.detach().cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref)), tag
        assert np.array_equal(np.isinf(got), np.isinf(ref)) and np.array_equal(got[np.isinf(ref)], ref[np.isinf(ref)]), tag
        fin = np.isfinite(ref)
        tol = 1e-12 if dt == np.float64 else 2e-6
        # fp32: the oracle's float build follows the reference's operation order, cancellations included (log k - k + k
        # at kappa = 700 leaves 1e-5 of rounding); the kernels follow it where the reference's own outputs need it and
        # use cancellation-free forms elsewhere, so a value must be close to the float restatement OR to the double one
        with np.errstate(all="ignore"):
            ref64, d64, _ = orc.vmf_kl(k.astype(np.float64), VARIANTS[name])
        # (the reference's forms subtract two numbers of size k -- k I1/I0 - log I0 in regular_cnn/s1_vae.py:83-88,
        #  -(k - ln 2) + k (1 + 2 e^-2k) in model/spherical_vae.py:172-182, -log sinh k + k coth k in the torus and
        #  circular forms: their float rounding is k * 2^-24 whoever evaluates them; the two stable variants are free of it)
        cond = np.abs(k[fin].astype(np.float64)) if (not name.startswith("stable") and dt == np.float32) else 0.0
        e = np.abs(got[fin] - ref[fin]) / (1 + np.abs(ref[fin]) + cond)
        ok64 = np.isfinite(ref64[fin])
        e64 = np.abs(got[fin] - ref64[fin]) / (1 + np.abs(ref64[fin]) + cond)
        e[ok64] = np.minimum(e[ok64], e64[ok64])
        key = "kl64" if dt == np.float64 else "kl32"
        if e.size:
            worst[key] = max(worst[key], float(e.max()))
            assert e.max() < tol, (tag, float(e.max()), float(k[fin][e.argmax()]))
        # gradient of the per-element values (upstream gradient 1)
        gfin = fin & np.isfinite(dref)
        if gfin.any():
            torch.where(torch.isfinite(out), out, torch.zeros_like(out)).sum().backward()
            dg = tk.grad.cpu().numpy()
            # the derivative is 1 / kappa minus a term of the same size below kappa ~ 1 (the reference's own autograd
            # subtracts them too): fp32 is held to the double-precision oracle relative to 1 + |d| + 1 / kappa
            gfin &= np.isfinite(d64)
            with np.errstate(all="ignore"):
                cond = 1 + np.abs(d64[gfin]) + 1 / np.abs(k[gfin].astype(np.float64))
            ed = np.minimum(np.abs(dg[gfin] - d64[gfin]), np.abs(dg[gfin] - dref[gfin])) / cond
            key = "dkl64" if dt == np.float64 else "dkl32"
            worst[key] = max(worst[key], float(ed.max()))
            assert ed.max() < (1e-11 if dt == np.float64 else 5e-6), (tag, float(ed.max()), float(k[gfin][ed.argmax()]))
        # fused sums (double accumulation) on the finite part
        kf = k[fin & (k > 0)] if name in ("s1", "stable_m2", "stable_m3", "torus", "circular") else k[fin]
        if kf.size:
            with np.errstate(all="ignore"):
                rf, _, sf = orc.vmf_kl(kf, VARIANTS[name])
            if np.isfinite(rf).all():
                tkf = _view(gen, kf)
                s_got = float(kl_sum(tkf, name))
                m_got = float(kl_mean(tkf, name))
                with np.errstate(all="ignore"):
                    _, _, sf64 = orc.vmf_kl(kf.astype(np.float64), VARIANTS[name])
                scale = max(1.0, float(np.abs(rf).sum()))
                if dt == np.float32 and not name.startswith("stable"):
                    scale += float(np.abs(kf.astype(np.float64)).sum())  # the same k * 2^-24 per element as above
                stol = 1e-12 if dt == np.float64 else 2e-6  # per-element rounding, same sign at worst
                es = min(abs(s_got - sf), abs(s_got - sf64) if np.isfinite(sf64) else np.inf) / scale
                worst["sum"] = max(worst["sum"], es)
                assert es < stol, (tag, es)
                em = min(abs(m_got - sf / kf.size), abs(m_got - sf64 / kf.size) if np.isfinite(sf64) else np.inf)
                assert em <= stol * scale / kf.size + 1e-30, (tag, em)
        # fused ELBO: x (rows_x, feat) against x_hat (reps, rows_x, feat) or equal shapes
        if case % 2 == 0:
            feat = int(gen.choice([1, 3, 7, 64, 784]))
            rows = int(gen.choice([1, 2, 5, 33, 128]))
            reps = int(gen.choice([0, 1, 3, 128 if rows * feat <= 128 * 64 else 2]))
            x = gen.uniform(0, 1, (rows, feat)).astype(dt)
            xh = gen.standard_normal(((reps, rows, feat) if reps else (rows, feat))).astype(dt)
            kk = np.exp(gen.uniform(0, np.log(100.0), rows)).astype(dt)
            gamma, beta = float(gen.uniform(0.1, 2)), float(gen.uniform(0.001, 1))
            ev = name if name in ("spherical", "torus", "circular") else "spherical"
            ref_l = co.elbo(x, xh, kk, gamma, beta, ev)
            tx, th, tkk = _view(gen, x), _view(gen, xh).requires_grad_(), _view(gen, kk).requires_grad_()
            loss = elbo_loss(tx, th, tkk, gamma, beta, ev)
            if not np.isfinite(float(ref_l)):  # the raw torus / circular forms overflow beyond kappa = 88.7 in float
                assert np.isnan(float(loss.detach())) == np.isnan(float(ref_l)), tag
                continue
            loss.backward()
            el = abs(float(loss.detach()) - float(ref_l)) / (1 + abs(float(ref_l)))
            worst["elbo"] = max(worst["elbo"], el)
            assert el < (1e-12 if dt == np.float64 else 2e-6), (tag, el)
            gref = -2 * gamma * (x.astype(np.float64) - xh.astype(np.float64)) / xh.size
            eg = np.abs(th.grad.cpu().numpy() - gref).max() / np.abs(gref).max()
            with np.errstate(all="ignore"):
                _, dkk, _ = orc.vmf_kl(kk.astype(np.float64), VARIANTS[ev])  # (the float build's 1/k - 1 + coth cancels)
            ek = np.abs(tkk.grad.cpu().numpy() - beta * dkk / rows).max() / (np.abs(beta * dkk / rows).max() + 1e-30)
            worst["delbo"] = max(worst["delbo"], float(max(eg, ek)))
            assert max(eg, ek) < (1e-11 if dt == np.float64 else 5e-6), (tag, eg, ek)
    return {"cases": cases, "worst": worst}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    print(sweep(a.cases, a.seed))
