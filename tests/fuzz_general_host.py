#!/usr/bin/env python3
"""Randomised sweep of the general-m host class (VonMisesFisher: rsample with and without the fused log_prob, sample
shapes, a shared mean, raw means normalised in the kernel, log_prob, entropy, kl_uniform, mean) on the GPU against
the CPU oracle, scipy and closed forms (test infrastructure; the GPU suite runs a short sweep through `sweep`):
random widths 2 .. 600, batch shapes, scale shapes, dtypes and concentrations, plus a moment check of a shared-mean
draw (E<z, mu> = A_m(kappa) and its variance) that does not depend on any restatement of the sampler.
python tests/fuzz_general_host.py [--cases 200] [--seed 0]"""
import argparse
import math
import os
import sys

import numpy as np
import scipy.special as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from medicalimageanalysis_amd import rng  # noqa: E402
from medicalimageanalysis_amd.sampler.von_mises_fisher import VonMisesFisher  # noqa: E402
from oracle import vmf_oracle as orc  # noqa: E402

DEV = torch.device("cuda:0")
WIDTHS = [2, 3, 4, 5, 6, 8, 10, 11, 16, 17, 24, 31, 32, 33, 50, 64, 65, 100, 128, 130, 200, 256, 300, 512, 600]


def _A(m, k):
    """mean resultant length A_m(kappa) = I_{m/2}(kappa) / I_{m/2-1}(kappa)"""
    k = np.asarray(k, dtype=np.float64)
    a = orc.bessel_ratio(m / 2, k.reshape(-1)).reshape(k.shape)  # finite where both scaled values underflow
    with np.errstate(all="ignore"):
        a_sp = sp.ive(m / 2, k) / sp.ive(m / 2 - 1, k)
    ok = np.isfinite(a_sp) & (sp.ive(m / 2, k) > 1e-280)
    assert np.allclose(a[ok], a_sp[ok], rtol=1e-11, atol=0), (m, k[ok], a[ok], a_sp[ok])
    return a


def sweep(cases, seed0):
    gen = np.random.default_rng(seed0)
    worst = {"logp64": 0.0, "logp32": 0.0, "fused_vs_separate": 0.0, "entropy": 0.0, "area": 0.0, "mean": 0.0,
             "moment_sigmas": 0.0}
    for case in range(cases):
        m = int(gen.choice(WIDTHS))
        dt = torch.float64 if gen.random() < 0.4 else torch.float32
        npdt = np.float64 if dt == torch.float64 else np.float32
        bshape = tuple(int(gen.choice([1, 2, 3, 7, 64, 129, 500])) for _ in range(int(gen.integers(1, 3))))
        nb = int(np.prod(bshape))
        ss = () if gen.random() < 0.6 else (int(gen.choice([1, 2, 5])),)
        kmax = 1e4 if dt == torch.float32 else 1e5
        kappa = np.exp(gen.uniform(np.log(0.05), np.log(kmax), bshape)).astype(npdt)
        mu = gen.standard_normal(bshape + (m,))
        mu /= np.linalg.norm(mu, axis=-1, keepdims=True)
        mu = mu.astype(npdt)
        normalize = bool(gen.random() < 0.4)
        raw_scale = np.exp(gen.uniform(-3, 3, bshape + (1,))).astype(npdt) if normalize else 1.0
        t_loc = torch.from_numpy((mu * raw_scale).astype(npdt)).to(DEV)
        k_shape = bshape + (1,) if gen.random() < 0.5 else bshape
        t_k = torch.from_numpy(kappa.reshape(k_shape)).to(DEV)
        tag = f"case {case}: m={m} {npdt.__name__} batch={bshape} sample={ss} scale={k_shape} normalize={normalize}"
        q = VonMisesFisher(t_loc, t_k, validate_args=False, normalize_loc=normalize,
                           generator=rng.PhiloxGenerator(int(gen.integers(1, 1 << 40)), int(gen.integers(0, 1 << 20))))
        z, logp = q.rsample(ss, return_log_prob=True)
        assert z.shape == ss + bshape + (m,) and logp.shape == ss + bshape, (tag, z.shape, logp.shape)
        zz = z.double()
        assert float((zz.norm(dim=-1) - 1).abs().max()) < (1e-9 if dt == torch.float64 else 2e-5), tag
        # fused log p(z) against the separate kernel and against the oracle (double, on the values the kernels saw)
        lp_sep = q.log_prob(z)
        assert lp_sep.shape == logp.shape, tag
        mu_unit = (mu.astype(np.float64) * raw_scale) if normalize else mu.astype(np.float64)
        mu_unit = mu_unit / np.linalg.norm(mu_unit, axis=-1, keepdims=True) if normalize else mu_unit
        S = int(np.prod(ss)) if ss else 1
        mu_rows = np.broadcast_to(mu_unit, ss + bshape + (m,)).reshape(S * nb, m)
        k_rows = np.broadcast_to(kappa.astype(np.float64), ss + bshape).reshape(S * nb)
        lp_ref = orc.vmf_log_prob(np.ascontiguousarray(mu_rows), np.ascontiguousarray(k_rows),
                                  zz.reshape(S * nb, m).cpu().numpy()).reshape(ss + bshape)
        # log p = kappa <mu, z> + log C_m(kappa): the float kernels round a sum of size kappa
        kb = np.broadcast_to(kappa.astype(np.float64), ss + bshape)
        cond = 1 + np.abs(lp_ref) + (kb if dt == torch.float32 else 1e-4 * kb)  # (double: 1e-11 * 1e-4 k = 4.5 ulp of k)
        e_f = float((np.abs(logp.double().cpu().numpy() - lp_ref) / cond).max())
        e_s = float((np.abs(lp_sep.double().cpu().numpy() - lp_ref) / cond).max())
        key = "logp64" if dt == torch.float64 else "logp32"
        worst[key] = max(worst[key], e_f, e_s)
        assert max(e_f, e_s) < (1e-11 if dt == torch.float64 else 1e-5), (tag, e_f, e_s)
        worst["fused_vs_separate"] = max(worst["fused_vs_separate"],
                                         float(((logp - lp_sep).double().abs().cpu().numpy() / cond).max()))
        # entropy, KL against the uniform law, their sum = log of the sphere's area, and the mean resultant length
        H, KL = q.entropy().double().cpu().numpy(), q.kl_uniform().double().cpu().numpy()
        assert H.shape == bshape and KL.shape == bshape, tag
        H_ref, KL_ref = orc.vmf_entropy_kl(kappa.astype(np.float64).reshape(-1), m)
        tol = 1e-10 if dt == torch.float64 else 2e-5
        e_h = float(np.abs(KL.reshape(-1) - KL_ref).max() / (1 + np.abs(KL_ref).max()))
        worst["entropy"] = max(worst["entropy"], e_h if dt == torch.float64 else 0.0)
        assert e_h < tol, (tag, e_h)
        area = math.log(2.0) + 0.5 * m * math.log(math.pi) - math.lgamma(0.5 * m)
        e_a = float(np.abs(H + KL - area).max() / (1 + abs(area)))
        worst["area"] = max(worst["area"], e_a if dt == torch.float64 else 0.0)
        assert e_a < tol, (tag, e_a)
        mean = q.mean.double().cpu().numpy()
        mean_ref = mu_unit * _A(m, kappa.astype(np.float64))[..., None]
        e_m = float(np.abs(mean - mean_ref).max())
        worst["mean"] = max(worst["mean"], e_m if dt == torch.float64 else 0.0)
        assert e_m < (1e-11 if dt == torch.float64 else 1e-5), (tag, e_m)
        # moments of a shared-mean draw (mu_stride = 0): w = <z, mu> has mean A and variance 1 - A^2 - (m-1) A / kappa
        if case % 4 == 0:
            k0 = float(np.exp(gen.uniform(np.log(0.3), np.log(3e3))))
            n = 40000
            q1 = VonMisesFisher(t_loc.reshape(-1, m)[:1], torch.tensor([k0], device=DEV, dtype=dt), validate_args=False,
                                normalize_loc=normalize, generator=rng.PhiloxGenerator(int(gen.integers(1, 1 << 40))))
            zs = q1.sample((n,)).double()
            assert zs.shape == (n, 1, m), (tag, zs.shape)
            mu0 = torch.from_numpy(mu_unit.reshape(-1, m)[0]).to(DEV)
            w = (zs[:, 0, :] * mu0).sum(-1)
            A = float(_A(m, k0))
            var = max(1 - A * A - (m - 1) * A / k0, 1e-12)
            sig = abs(float(w.mean()) - A) / math.sqrt(var / n)
            worst["moment_sigmas"] = max(worst["moment_sigmas"], sig)
            assert sig < 5.0, (tag, k0, float(w.mean()), A, sig)
            assert abs(float(w.var()) - var) < 0.1 * var + 5e-6, (tag, k0, float(w.var()), var)
            # the component orthogonal to mu is isotropic: its mean vanishes
            perp = zs[:, 0, :] - w[:, None] * mu0
            assert float(perp.mean(0).abs().max()) < 5.0 * math.sqrt(max(1 - A * A, 1e-12) / max(m - 1, 1) / n) + 1e-6, tag
    return {"cases": cases, "worst": worst}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    print(sweep(a.cases, a.seed))
