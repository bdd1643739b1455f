"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the steps on either side of the sampler in the reference's train
step.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product never does.

Each function follows the cited reference lines literally (same operation order, same dtype), so that fp32 results
agree with torch to rounding.  Pinned by tests/golden/g_consumers.npz: `_elbo` values come from executing the
reference's SphericalVAE / CircularVAE; `latent_loss` and `_build_torus` live in modules that import geomstats (not
installed), so their golden values come from the same formulas evaluated with torch ops by tests/golden/make_golden.py
(gradients by torch autograd) -- parity for those two is pinned by restated formulas, not by reference execution.
"""
import numpy as np


def _kl(kappa, variant):
    """per-element KL; variants as in model/spherical_vae.py:141-184, model/torus_vae.py:182-194,
    CNN_experiment/circular_VAE.py:118-130"""
    k = np.asarray(kappa)
    dt = k.dtype.type
    with np.errstate(all="ignore"):
        if variant == "spherical":
            kk = np.maximum(k, dt(1e-8))
            out = np.zeros_like(kk)
            small, large = kk < dt(1e-3), kk > dt(20.0)
            mid = ~(small | large)
            km = kk[mid]
            sh = np.sinh(km) + dt(1e-8)
            out[mid] = np.log(km) - np.log(sh) + km * (np.cosh(km) / sh) - dt(1.0)
            kl = kk[large]
            out[large] = np.log(kl) - (kl - dt(np.log(2.0))) + kl * (dt(1.0) + dt(2.0) * np.exp(dt(-2.0) * kl)) - dt(1.0)
            return out
        if variant == "torus":
            sh = np.sinh(k)
            return np.log(k) - np.log(sh) + k * (np.cosh(k) / sh) - dt(1.0)
        if variant == "circular":
            sh = np.sinh(k)
            return np.log(k) - np.log(sh + dt(1e-6)) + k * (np.cosh(k) / (sh + dt(1e-6))) - dt(1.0)
    raise ValueError(variant)


def elbo(x, x_hat, kappa, gamma, beta, variant="spherical"):
    """model/spherical_vae.py:133-138: gamma * mean((x - x_mu)^2) + beta * mean(KL); x broadcasts against x_hat.
    Sums are carried in double and each term is rounded to the input dtype before they are combined."""
    dt = x_hat.dtype.type
    d = (x - x_hat).astype(np.float64)
    rec = dt(np.mean(d * d))
    kl = dt(np.mean(_kl(kappa, variant).astype(np.float64)))
    return dt(dt(gamma) * rec + dt(beta) * kl)


def build_torus(z_theta, z_phi, R, r):
    """model/torus_vae.py:60-81 (y uses r + r cos theta, :78)"""
    dt = z_theta.dtype.type
    ct, st = z_theta[..., 0], z_theta[..., 1]
    cp, sp = z_phi[..., 0], z_phi[..., 1]
    return np.stack([(dt(R) + dt(r) * ct) * cp, (dt(r) + dt(r) * ct) * sp, dt(r) * st], axis=-1)


def build_torus_bwd(z_theta, z_phi, g, R, r):
    dt = z_theta.dtype.type
    ct = z_theta[..., 0]
    cp, sp = z_phi[..., 0], z_phi[..., 1]
    gt, gp = np.zeros_like(z_theta), np.zeros_like(z_phi)
    gt[..., 0] = dt(r) * (cp * g[..., 0] + sp * g[..., 1])
    gt[..., 1] = dt(r) * g[..., 2]
    gp[..., 0] = (dt(R) + dt(r) * ct) * g[..., 0]
    gp[..., 1] = (dt(r) + dt(r) * ct) * g[..., 1]
    return gt, gp


def _wrap(a):
    two_pi = a.dtype.type(2 * np.pi)
    return np.mod(a + two_pi, two_pi)          # numpy's mod is the floored modulo, like torch's %


def latent_loss_rows(labels, z, dataset, R=0.0, r=0.0):
    """per-row values whose mean is dataloader/utils.py:101-180's return value"""
    dt = z.dtype.type
    labels = labels.astype(z.dtype)
    if dataset == "S1_dataset":                                  # :113-123
        phi = _wrap(np.arctan2(z[..., 1], z[..., 0]))
        return (dt(1) - np.cos(phi - labels[:, 0])) ** 2
    if dataset == "S2_dataset":                                  # :128-151
        zn = z / np.linalg.norm(z, axis=-1, keepdims=True)
        zc = np.clip(zn[..., 2], dt(-1), dt(1))
        th = np.arccos(zc)
        ph = _wrap(np.arctan2(z[..., 1], z[..., 0]))
        dth, dph = labels[:, 0] - th, labels[:, 1] - ph
        t1 = dt(1) - np.cos(dth)
        t2 = np.sin(labels[:, 0]) * np.sin(th) * (dt(1) - np.cos(dph))
        return (t1 + t2) ** 2
    if dataset == "T2_dataset":                                  # :153-177
        x, y, zc = z[..., 0], z[..., 1], z[..., 2]
        rho = np.sqrt(x * x + y * y)
        ph = _wrap(np.arctan2(y, x))
        th = _wrap(np.arctan2(zc, rho - dt(R)))
        q = (rho - dt(R)) ** 2 + zc * zc - dt(r) * dt(r)
        return (dt(1) - np.cos(th - labels[:, 0])) + (dt(1) - np.cos(ph - labels[:, 1])) + dt(0.1) * q * q
    raise ValueError("Unknown dataset type in latent_loss()")


def latent_loss(labels, z, dataset, R=0.0, r=0.0):
    rows = latent_loss_rows(labels, z, dataset, R, r)
    return z.dtype.type(np.mean(rows.astype(np.float64)))


def latent_loss_bwd(labels, z, dataset, R=0.0, r=0.0, h=1e-6):
    """d mean / d z by central differences in double (an independent check of the analytic device gradient)"""
    z64 = z.astype(np.float64)
    l64 = labels.astype(np.float64)
    g = np.zeros_like(z64)
    n = z64.size // z64.shape[-1]
    for c in range(z64.shape[-1]):
        zp, zm = z64.copy(), z64.copy()
        zp[..., c] += h
        zm[..., c] -= h
        g[..., c] = (latent_loss_rows(l64, zp, dataset, R, r) - latent_loss_rows(l64, zm, dataset, R, r)) / (2 * h) / n
    return g
