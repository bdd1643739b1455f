#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the reference.

Run ONLY in the build container (it needs /root/reference and scipy):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference's modules unmodified, feeds them seeded inputs,
captures the noise they draw (by replaying the torch generator in the same
order the reference draws: ``rand`` for ``u`` then ``randn`` for ``eps``) and
stores inputs + outputs as small ``.npz`` files.  Nothing of the reference's
source is stored: only arrays.  The tests and the GPU box read the ``.npz``
files and never the reference.

Files written (all float arrays keep the dtype the reference computed in):
  g_rsample.npz   VonMisesFisher3D.rsample   sampler/von_mises_fisher.py:50-86
  g_grads.npz     autograd through rsample   sampler/von_mises_fisher.py:66-86
  g_kl.npz        the four KL variants       model/spherical_vae.py:141-184,
                                             CNN_experiment/circular_VAE.py:118-130,
                                             regular_cnn/s1_vae.py:70-92,
                                             (torus: formula of model/torus_vae.py:190-194,
                                              evaluated with torch; the module itself needs
                                              geomstats, which is not installed)
  g_ive.npz       IveFunction fwd/bwd        sampler/ive.py:12-63 (scipy 1.15.3)
  g_logprob.npz   log_prob/_log_normalization CNN_experiment/von_mises_fisher.py:88-108
  g_mean.npz      VonMisesFisher3D.mean      sampler/von_mises_fisher.py:38-43
  g_uniform.npz   SphericalUniform.sample    sampler/spherical_uniform.py:25-33
  g_svae.npz      SphericalVAE fwd/_elbo/bwd model/spherical_vae.py:51-138
  g_cvae.npz      CircularVAE fwd/_elbo      CNN_experiment/circular_VAE.py:42-116
  g_s1.npz        S1_VAE reparam + loss      regular_cnn/s1_vae.py:44-92
  g_consumers.npz _elbo on given tensors (executed: SphericalVAE._elbo model/spherical_vae.py:119-138,
                  CircularVAE._elbo CNN_experiment/circular_VAE.py:90-116), with gradients; _build_torus
                  (model/torus_vae.py:60-81) and latent_loss (dataloader/utils.py:101-180) by their formulas
                  evaluated with torch ops + autograd (their modules import geomstats, not installed)
  g_curvature.npz compute_curvature_profile  regular_cnn/geometry.py:5-64 on a seeded S1_VAE
"""
import importlib
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def _purge(*names):
    for n in list(sys.modules):
        if n in names or any(n.startswith(p + ".") for p in names):
            del sys.modules[n]


def _import_from(path, module):
    """Import `module` with `path` first on sys.path (reference dirs shadow each other)."""
    _purge("sampler", "model", "von_mises_fisher", "ive", "circular_VAE", "s1_vae")
    sys.path.insert(0, path)
    try:
        return importlib.import_module(module)
    finally:
        sys.path.pop(0)


def np_(t):
    return t.detach().cpu().numpy()


def kappa_grid(B, dtype, gen):
    edge = torch.tensor([1e-3, 0.5, 1.0, 20.0, 100.0, 1e3, 1e4], dtype=dtype)
    k = torch.exp(torch.rand(B - edge.numel(), generator=gen, dtype=dtype) * np.log(1e4))
    return torch.cat([edge, k])


def make_rsample():
    vm = _import_from(REF, "sampler.von_mises_fisher")
    out = {}
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        for seed in range(4):
            gen = torch.Generator().manual_seed(1000 + seed)
            # ---- elementwise: scale (B,)
            B = 257
            loc = torch.randn(B, 3, generator=gen, dtype=dt)
            scale = kappa_grid(B, dt, gen)
            torch.manual_seed(seed)
            u = torch.rand(scale.shape, dtype=dt)
            eps = torch.randn(loc.shape, dtype=dt)
            torch.manual_seed(seed)
            q = vm.VonMisesFisher3D(loc, scale)
            z = q.rsample()
            torch.manual_seed(seed)
            w = q._sample_w3(torch.Size())
            assert torch.equal(
                w, 1 + torch.logsumexp(torch.stack([u.log(), (1 - u).log() - 2 * scale]), 0) / scale
            )
            p = f"ew_{tag}_s{seed}_"
            out.update({p + "loc": np_(loc), p + "scale": np_(scale), p + "u": np_(u),
                        p + "eps": np_(eps), p + "w": np_(w), p + "z": np_(z),
                        p + "locn": np_(q.loc)})
            # ---- outer: scale (B,1)  -> z (B,B,3)
            B = 33
            loc = torch.randn(B, 3, generator=gen, dtype=dt)
            scale = kappa_grid(B, dt, gen).unsqueeze(-1)
            torch.manual_seed(seed)
            u = torch.rand(scale.shape, dtype=dt)
            eps = torch.randn(loc.shape, dtype=dt)
            torch.manual_seed(seed)
            z = vm.VonMisesFisher3D(loc, scale).rsample()
            assert z.shape == (B, B, 3)
            p = f"outer_{tag}_s{seed}_"
            out.update({p + "loc": np_(loc), p + "scale": np_(scale), p + "u": np_(u),
                        p + "eps": np_(eps), p + "z": np_(z)})
        # ---- sample_shape (5,) with scale (B,)
        gen = torch.Generator().manual_seed(77)
        B = 19
        loc = torch.randn(B, 3, generator=gen, dtype=dt)
        scale = kappa_grid(B, dt, gen)
        torch.manual_seed(5)
        u = torch.rand((5,) + scale.shape, dtype=dt)
        eps = torch.randn((5,) + loc.shape, dtype=dt)
        torch.manual_seed(5)
        z = vm.VonMisesFisher3D(loc, scale).rsample(torch.Size([5]))
        assert z.shape == (5, B, 3)
        p = f"ss5_{tag}_"
        out.update({p + "loc": np_(loc), p + "scale": np_(scale), p + "u": np_(u),
                    p + "eps": np_(eps), p + "z": np_(z)})
        # ---- sample_shape (4,) with scale (4,1): the (S,Bs,1,1)x(S,Bl,3) broadcast
        B = 4
        loc = torch.randn(B, 3, generator=gen, dtype=dt)
        scale = kappa_grid(11, dt, gen)[7:11].unsqueeze(-1)
        torch.manual_seed(6)
        u = torch.rand((4,) + scale.shape, dtype=dt)
        eps = torch.randn((4,) + loc.shape, dtype=dt)
        torch.manual_seed(6)
        z = vm.VonMisesFisher3D(loc, scale).rsample(torch.Size([4]))
        p = f"ss4o_{tag}_"
        out.update({p + "loc": np_(loc), p + "scale": np_(scale), p + "u": np_(u),
                    p + "eps": np_(eps), p + "z": np_(z)})
        # ---- batch dims (2,3) with scale (2,3)
        loc = torch.randn(2, 3, 3, generator=gen, dtype=dt)
        scale = torch.rand(2, 3, generator=gen, dtype=dt) * 50 + 0.1
        torch.manual_seed(7)
        u = torch.rand(scale.shape, dtype=dt)
        eps = torch.randn(loc.shape, dtype=dt)
        torch.manual_seed(7)
        z = vm.VonMisesFisher3D(loc, scale).rsample()
        p = f"nd_{tag}_"
        out.update({p + "loc": np_(loc), p + "scale": np_(scale), p + "u": np_(u),
                    p + "eps": np_(eps), p + "z": np_(z)})
    np.savez_compressed(os.path.join(OUT, "g_rsample.npz"), **out)


def make_grads():
    vm = _import_from(REF, "sampler.von_mises_fisher")
    out = {}
    dt = torch.float64
    gen = torch.Generator().manual_seed(4242)
    for name, B, outer in (("ew", 65, False), ("outer", 17, True)):
        loc = torch.randn(B, 3, generator=gen, dtype=dt).requires_grad_()
        scale = kappa_grid(B, dt, gen)
        if outer:
            scale = scale.unsqueeze(-1)
        scale = scale.clone().requires_grad_()
        torch.manual_seed(11)
        u = torch.rand(scale.shape, dtype=dt)
        eps = torch.randn(loc.shape, dtype=dt)
        torch.manual_seed(11)
        z = vm.VonMisesFisher3D(loc, scale).rsample()
        c = torch.randn(z.shape, generator=gen, dtype=dt)
        (z * c).sum().backward()
        p = name + "_"
        out.update({p + "loc": np_(loc), p + "scale": np_(scale), p + "u": np_(u), p + "eps": np_(eps),
                    p + "z": np_(z), p + "c": np_(c), p + "dloc": np_(loc.grad), p + "dscale": np_(scale.grad)})
    np.savez_compressed(os.path.join(OUT, "g_grads.npz"), **out)


def make_kl():
    sv = _import_from(REF, "model.spherical_vae")
    cfg = dict(extrinsic_dim=3, latent_dim=3, sftbeta=1.0, encoder_width=8, encoder_depth=1,
               decoder_width=8, decoder_depth=1, dropout_p=0.1, device="cpu", gamma=1.0, beta=0.01)
    svae = sv.SphericalVAE(cfg)
    cv = _import_from(os.path.join(REF, "CNN_experiment"), "circular_VAE")
    cvae = cv.CircularVAE(dict(device="cpu", extrinsic_dim=4, encoder_width=8, decoder_width=8))
    s1 = _import_from(os.path.join(REF, "regular_cnn"), "s1_vae")
    out = {}
    base = np.concatenate([
        [1e-9, 1e-8, 1e-4, 9.99e-4, 1e-3, 1.001e-3, 0.5, 1.0, 5.0, 19.9, 20.0, 20.1, 50.0, 80.0, 88.0, 88.7,
         88.8, 89.4, 89.5, 100.0, 500.0, 709.0, 710.0, 711.0, 800.0, 1e3, 1e4, 1e5],
        np.logspace(-2, 4, 61),
    ])
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        k = torch.tensor(base, dtype=dt)
        out[f"kappa_{tag}"] = np_(k)
        out[f"spherical_{tag}"] = np_(svae.kl_vmf_spherical_uniform(k))
        out[f"circular_{tag}"] = np_(cvae.kl_vmf_spherical_uniform(k))
        # torus: one-line formula of model/torus_vae.py:190-194 evaluated with torch
        sinh_k = torch.sinh(k)
        out[f"torus_{tag}"] = np_(torch.log(k) - torch.log(sinh_k) + k * (torch.cosh(k) / sinh_k) - 1.0)
        # S1: per-point terms (regular_cnn/s1_vae.py:83-88) and the summed loss with recon==x
        i0, i1 = torch.special.i0(k), torch.special.i1(k)
        out[f"s1_{tag}"] = np_(k * (i1 / i0) - torch.log(i0))
        kk = torch.tensor([1.0, 2.5, 10.0, 40.0], dtype=dt).unsqueeze(-1)
        x = torch.zeros(4, 2, dtype=dt)
        out[f"s1_loss_kappa_{tag}"] = np_(kk)
        out[f"s1_loss_{tag}"] = np_(s1.loss_function(x + 0.5, x, None, kk))
    # gradient of the spherical piecewise KL (autograd through the masked writes)
    k = torch.tensor(base, dtype=torch.float64, requires_grad=True)
    svae.kl_vmf_spherical_uniform(k).sum().backward()
    out["spherical_grad_f64"] = np_(k.grad)
    np.savez_compressed(os.path.join(OUT, "g_kl.npz"), **out)


def make_ive():
    iv = _import_from(REF, "sampler.ive")
    out = {}
    orders = [0, 1, 0.5, 1.5, 4, 4.5, 9, 31, 32, 63.5, 255, 256]
    zs = np.concatenate([np.logspace(-3, 5, 97), [0.0, 1e-9, 2.0, 30.0, 1e300]])
    neg = np.array([-1.0, -0.25, -30.0])
    out["orders"] = np.array(orders, dtype=np.float64)
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        zz = np.concatenate([zs[:-1] if tag == "f32" else zs, neg])
        out[f"z_{tag}"] = zz.astype(np.float64 if tag == "f64" else np.float32)
        for v in orders:
            z = torch.tensor(zz, dtype=dt, requires_grad=True)
            y = iv.ive(v, z)
            (g,) = torch.autograd.grad(y.sum(), z)
            out[f"fwd_{tag}_v{v}"] = np_(y)
            out[f"bwd_{tag}_v{v}"] = np_(g)
    # module wrapper gives the same numbers
    z = torch.tensor([0.3, 3.0, 300.0], dtype=torch.float64)
    assert torch.equal(iv.Ive(2.5)(z), iv.ive(2.5, z))
    np.savez_compressed(os.path.join(OUT, "g_ive.npz"), **out)


def make_logprob_mean_uniform():
    cvm = _import_from(os.path.join(REF, "CNN_experiment"), "von_mises_fisher")
    out = {}
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        gen = torch.Generator().manual_seed(99)
        k = torch.tensor([0.5, 1.0, 10.0, 50.0, 88.0, 89.5, 100.0, 700.0, 800.0], dtype=dt)
        loc = torch.randn(k.numel(), 3, generator=gen, dtype=dt)
        x = torch.randn(k.numel(), 3, generator=gen, dtype=dt)
        x = x / x.norm(dim=-1, keepdim=True)
        q = cvm.VonMisesFisher3D(loc, k)
        out.update({f"loc_{tag}": np_(loc), f"scale_{tag}": np_(k), f"x_{tag}": np_(x),
                    f"logprob_{tag}": np_(q.log_prob(x)), f"lognorm_{tag}": np_(q._log_normalization())})
    np.savez_compressed(os.path.join(OUT, "g_logprob.npz"), **out)

    vm = _import_from(REF, "sampler.von_mises_fisher")
    out = {}
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        gen = torch.Generator().manual_seed(98)
        k = torch.tensor([1e-2, 0.5, 1.0, 10.0, 50.0, 100.0, 1e3, 1e4], dtype=dt)
        loc = torch.randn(k.numel(), 3, generator=gen, dtype=dt)
        q = vm.VonMisesFisher3D(loc, k)
        out.update({f"loc_{tag}": np_(loc), f"scale_{tag}": np_(k), f"mean_{tag}": np_(q.mean()),
                    f"stddev_{tag}": np_(q.stddev())})
    np.savez_compressed(os.path.join(OUT, "g_mean.npz"), **out)

    su = _import_from(REF, "sampler.spherical_uniform")
    out = {}
    for dim in (2, 3, 10):
        torch.manual_seed(3)
        g = torch.randn((6, dim))
        torch.manual_seed(3)
        s = su.SphericalUniform(dim).sample(torch.Size([6]))
        out[f"randn_d{dim}"] = np_(g)
        out[f"sample_d{dim}"] = np_(s)
    np.savez_compressed(os.path.join(OUT, "g_uniform.npz"), **out)


def make_svae():
    sv = _import_from(REF, "model.spherical_vae")
    cfg = dict(extrinsic_dim=5, latent_dim=3, sftbeta=1.0, encoder_width=16, encoder_depth=2,
               decoder_width=16, decoder_depth=2, dropout_p=0.1, device="cpu", gamma=1.0, beta=0.01)
    torch.manual_seed(0)
    model = sv.SphericalVAE(cfg).double()
    B = 8
    gen = torch.Generator().manual_seed(123)
    x = torch.randn(B, cfg["extrinsic_dim"], generator=gen, dtype=torch.float64)
    torch.manual_seed(21)
    u = torch.rand((B, 1), dtype=torch.float64)
    eps = torch.randn((B, 3), dtype=torch.float64)
    torch.manual_seed(21)
    z, x_mu, (z_mu, z_kappa) = model(x)
    assert z.shape == (B, B, 3) and x_mu.shape == (B, B, cfg["extrinsic_dim"])
    loss = model._elbo(x, x_mu, (z_mu, z_kappa))
    loss.backward()
    out = {"x": np_(x), "u": np_(u), "eps": np_(eps), "z": np_(z), "x_mu": np_(x_mu), "z_mu": np_(z_mu),
           "z_kappa": np_(z_kappa), "loss": np_(loss)}
    for k, v in model.state_dict().items():
        out["w::" + k] = np_(v)
    for k, v in model.named_parameters():
        out["g::" + k] = np_(v.grad)
    for k, v in cfg.items():
        if k != "device":
            out["cfg::" + k] = np.array(v)
    np.savez_compressed(os.path.join(OUT, "g_svae.npz"), **out)


def make_cvae():
    cv = _import_from(os.path.join(REF, "CNN_experiment"), "circular_VAE")
    cfg = dict(device="cpu", extrinsic_dim=6, encoder_width=12, decoder_width=12, gamma=1.0, beta=0.1)
    torch.manual_seed(1)
    model = cv.CircularVAE(cfg).double()
    B = 7
    gen = torch.Generator().manual_seed(321)
    x = torch.randn(B, cfg["extrinsic_dim"], generator=gen, dtype=torch.float64)
    torch.manual_seed(22)
    u = torch.rand((B,), dtype=torch.float64)
    eps = torch.randn((B, 3), dtype=torch.float64)
    torch.manual_seed(22)
    z, x_recon, (mu, kappa) = model(x)
    assert z.shape == (B, 3)
    loss = model._elbo(x, x_recon, (mu, kappa))
    loss.backward()
    out = {"x": np_(x), "u": np_(u), "eps": np_(eps), "z": np_(z), "x_recon": np_(x_recon), "mu": np_(mu),
           "kappa": np_(kappa), "loss": np_(loss)}
    for k, v in model.state_dict().items():
        out["w::" + k] = np_(v)
    for k, v in model.named_parameters():
        out["g::" + k] = np_(v.grad)
    for k, v in cfg.items():
        if k != "device":
            out["cfg::" + k] = np.array(v)
    np.savez_compressed(os.path.join(OUT, "g_cvae.npz"), **out)


def make_s1():
    s1 = _import_from(os.path.join(REF, "regular_cnn"), "s1_vae")
    torch.manual_seed(2)
    model = s1.S1_VAE(input_dim=6, hidden_dim=10).double()
    B = 9
    gen = torch.Generator().manual_seed(55)
    x = torch.randn(B, 6, generator=gen, dtype=torch.float64)
    mean_vec, kappa = model.encode(x)
    torch.manual_seed(23)
    noise = torch.randn_like(mean_vec)
    torch.manual_seed(23)
    z = model.reparameterize(mean_vec, kappa)
    torch.manual_seed(23)
    x_recon, mean_vec2, kappa2, z2 = model(x)
    assert torch.equal(z, z2)
    loss = s1.loss_function(x_recon, x, mean_vec2, kappa2)
    loss.backward()
    out = {"x": np_(x), "noise": np_(noise), "mean_vec": np_(mean_vec), "kappa": np_(kappa), "z": np_(z),
           "x_recon": np_(x_recon), "loss": np_(loss)}
    for k, v in model.state_dict().items():
        out["w::" + k] = np_(v)
    for k, v in model.named_parameters():
        out["g::" + k] = np_(v.grad)
    # standalone sampler case in both dtypes, large kappa included
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        gen = torch.Generator().manual_seed(56)
        m = torch.nn.functional.normalize(torch.randn(31, 2, generator=gen, dtype=dt), dim=1)
        k = torch.exp(torch.rand(31, 1, generator=gen, dtype=dt) * np.log(1e3))
        torch.manual_seed(24)
        nz = torch.randn_like(m)
        torch.manual_seed(24)
        zz = model.reparameterize(m, k)
        out.update({f"pn_mean_{tag}": np_(m), f"pn_kappa_{tag}": np_(k), f"pn_noise_{tag}": np_(nz),
                    f"pn_z_{tag}": np_(zz)})
    np.savez_compressed(os.path.join(OUT, "g_s1.npz"), **out)


def make_curvature():
    """regular_cnn/geometry.py:5-64 executed on a seeded S1_VAE (weights stored, so the test rebuilds the model)"""
    s1 = _import_from(os.path.join(REF, "regular_cnn"), "s1_vae")
    sys.path.insert(0, os.path.join(REF, "regular_cnn"))
    try:
        geo = importlib.import_module("geometry")
    finally:
        sys.path.pop(0)
    out = {}
    # fp32 only: the reference's thetas are torch.linspace's default dtype, so a double model raises there
    for dt, tag in ((torch.float32, "f32"),):
        torch.manual_seed(5)
        model = s1.S1_VAE(input_dim=7, hidden_dim=12).to(dt)
        thetas, prof = geo.compute_curvature_profile(model, "cpu", num_points=45)
        out[f"thetas_{tag}"], out[f"profile_{tag}"] = np.asarray(thetas), np.asarray(prof)
        for k, v in model.state_dict().items():
            out[f"w_{tag}::" + k] = np_(v)
    np.savez_compressed(os.path.join(OUT, "g_curvature.npz"), **out)


def _latent_loss_formula(labels, z, dataset, R=None, r=None):
    """dataloader/utils.py:101-180 evaluated with the same torch operations, in the same order"""
    if dataset == "S1_dataset":
        latent_phi = torch.atan2(z[..., 1], z[..., 0])
        latent_phi = (latent_phi + (2 * torch.pi)) % (2 * torch.pi)
        diff = torch.cos(latent_phi - labels[:, 0])
        return ((1 - diff) ** 2).mean()
    if dataset == "S2_dataset":
        theta_gt, phi_gt = labels[:, 0], labels[:, 1]
        z_norm = z / (z.norm(dim=-1, keepdim=True))
        x, y = z[..., 0], z[..., 1]
        z_comp = z_norm[..., 2].clamp(-1.0, 1.0)
        theta_hat = torch.acos(z_comp)
        phi_hat = torch.atan2(y, x)
        phi_hat = (phi_hat + (2 * torch.pi)) % (2 * torch.pi)
        dtheta, dphi = theta_gt - theta_hat, phi_gt - phi_hat
        term1 = 1.0 - torch.cos(dtheta)
        term2 = torch.sin(theta_gt) * torch.sin(theta_hat) * (1.0 - torch.cos(dphi))
        return ((term1 + term2) ** 2).mean()
    x, y, zc = z[..., 0], z[..., 1], z[..., 2]
    rho = torch.sqrt(x ** 2 + y ** 2)
    latent_phis = torch.atan2(y, x)
    latent_thetas = torch.atan2(zc, rho - R)
    latent_phis = (latent_phis + (2 * torch.pi)) % (2 * torch.pi)
    latent_thetas = (latent_thetas + (2 * torch.pi)) % (2 * torch.pi)
    thetas_loss = torch.mean(1 - torch.cos(latent_thetas - labels[:, 0]))
    phis_loss = torch.mean(1 - torch.cos(latent_phis - labels[:, 1]))
    torus_constraint = ((rho - R) ** 2 + zc ** 2 - r ** 2) ** 2
    return (thetas_loss + phis_loss + 0.1 * torus_constraint).mean()


def make_consumers():
    out = {}
    sv = _import_from(REF, "model.spherical_vae")
    cfg = dict(extrinsic_dim=12, latent_dim=3, sftbeta=1.0, encoder_width=8, encoder_depth=1, decoder_width=8,
               decoder_depth=1, dropout_p=0.0, device="cpu", gamma=0.7, beta=0.03)
    torch.manual_seed(0)
    smodel = sv.SphericalVAE(cfg)
    cv = _import_from(os.path.join(REF, "CNN_experiment"), "circular_VAE")
    ccfg = dict(device="cpu", extrinsic_dim=12, encoder_width=8, decoder_width=8, gamma=1.3, beta=0.2)
    cmodel = cv.CircularVAE(ccfg)
    gen = torch.Generator().manual_seed(77)
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        B, D = 9, 12
        x = torch.randn(B, D, generator=gen, dtype=dt)
        z_mu = torch.randn(B, 3, generator=gen, dtype=dt)
        # spherical: x_mu (B, B, D), kappa (B, 1) spanning the three KL regions
        x_mu = torch.randn(B, B, D, generator=gen, dtype=dt).requires_grad_()
        kap = torch.tensor([1e-4, 0.5, 1.0, 5.0, 19.9, 20.1, 50.0, 100.0, 1e3], dtype=dt).reshape(B, 1).requires_grad_()
        loss = smodel._elbo(x, x_mu, (z_mu, kap))
        loss.backward()
        out[f"sph_{tag}::x"], out[f"sph_{tag}::x_mu"], out[f"sph_{tag}::kappa"] = np_(x), np_(x_mu), np_(kap)
        out[f"sph_{tag}::loss"], out[f"sph_{tag}::g_x_mu"], out[f"sph_{tag}::g_kappa"] = np_(loss), np_(x_mu.grad), np_(kap.grad)
        # circular: x_recon (B, D), kappa (B,)
        x_rec = torch.randn(B, D, generator=gen, dtype=dt).requires_grad_()
        kap2 = (torch.rand(B, generator=gen, dtype=dt) * 40 + 1).requires_grad_()
        loss = cmodel._elbo(x, x_rec, (z_mu, kap2))
        loss.backward()
        out[f"cir_{tag}::x"], out[f"cir_{tag}::x_mu"], out[f"cir_{tag}::kappa"] = np_(x), np_(x_rec), np_(kap2)
        out[f"cir_{tag}::loss"], out[f"cir_{tag}::g_x_mu"], out[f"cir_{tag}::g_kappa"] = np_(loss), np_(x_rec.grad), np_(kap2.grad)
        # torus embedding (model/torus_vae.py:60-81) on (B, B, 3) sphere draws and on raw (B, 3) means
        R, r = 1.5, 0.5
        for name, shape in (("bb", (B, B, 3)), ("b", (B, 3))):
            zt = torch.nn.functional.normalize(torch.randn(*shape, generator=gen, dtype=dt), dim=-1).requires_grad_()
            zp = torch.nn.functional.normalize(torch.randn(*shape, generator=gen, dtype=dt), dim=-1).requires_grad_()
            ct, st, cp, sp = zt[..., 0], zt[..., 1], zp[..., 0], zp[..., 1]
            tor = torch.stack([(R + r * ct) * cp, (r + r * ct) * sp, r * st], dim=-1)
            c = torch.randn(*tor.shape, generator=gen, dtype=dt)
            (tor * c).sum().backward()
            k = f"tor_{name}_{tag}::"
            out[k + "zt"], out[k + "zp"], out[k + "out"], out[k + "c"] = np_(zt), np_(zp), np_(tor), np_(c)
            out[k + "g_zt"], out[k + "g_zp"] = np_(zt.grad), np_(zp.grad)
        # latent losses: z (B, B, 3) (the broadcast sample) and (B, 3); T2 on points near the torus
        for ds in ("S1_dataset", "S2_dataset", "T2_dataset"):
            for name, shape in (("bb", (B, B, 3)), ("b", (B, 3))):
                labels = torch.rand(B, 2, generator=gen, dtype=dt) * torch.tensor([np.pi, 2 * np.pi], dtype=dt)
                if ds == "T2_dataset":
                    labels = torch.rand(B, 2, generator=gen, dtype=dt) * (2 * np.pi)
                    th = torch.rand(*shape[:-1], generator=gen, dtype=dt) * (2 * np.pi)
                    ph = torch.rand(*shape[:-1], generator=gen, dtype=dt) * (2 * np.pi)
                    z = torch.stack([(R + r * th.cos()) * ph.cos(), (R + r * th.cos()) * ph.sin(), r * th.sin()], -1)
                    z = z + 0.05 * torch.randn(*shape, generator=gen, dtype=dt)
                else:
                    z = torch.nn.functional.normalize(torch.randn(*shape, generator=gen, dtype=dt), dim=-1)
                    z = z * (1 + 0.01 * torch.randn(*shape[:-1], 1, generator=gen, dtype=dt))
                z = z.clone().requires_grad_()
                val = _latent_loss_formula(labels, z, ds, R, r)
                val.backward()
                k = f"ll_{ds[:2]}_{name}_{tag}::"
                out[k + "labels"], out[k + "z"], out[k + "loss"], out[k + "g_z"] = np_(labels), np_(z), np_(val), np_(z.grad)
    np.savez_compressed(os.path.join(OUT, "g_consumers.npz"), **out)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not present: golden vectors can only be regenerated in the build container")
    torch.set_num_threads(1)
    if sys.argv[1:] == ["consumers"]:
        make_consumers()
        sys.exit(0)
    if sys.argv[1:] == ["curvature"]:
        make_curvature()
        sys.exit(0)
    make_rsample()
    make_grads()
    make_kl()
    make_ive()
    make_logprob_mean_uniform()
    make_svae()
    make_cvae()
    make_s1()
    make_consumers()
    make_curvature()
    total = 0
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            sz = os.path.getsize(os.path.join(OUT, f))
            total += sz
            print(f"{f:20s} {sz:8d} B")
    print("total", total)
