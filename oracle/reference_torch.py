"""CPU restatement, in plain torch ops, of the reference's m = 3 latent path: what the reference itself executes on the
host, one small torch op after another.  TEST INFRASTRUCTURE: only tests/ and the cpu_baseline leg of bench.py load it
(it is the thing timed next to the GPU, never a fallback of the product).  Pinned by tests/test_oracle_golden.py
against the vectors the reference produced (g_rsample, g_kl, g_svae).

  rsample          sampler/von_mises_fisher.py:27-35 (loc / (|loc| + 1e-5)), :50-64 (w by inverse CDF),
                   :66-86 (tangent, combine; a (B, 1) scale broadcasts the result to (B, B, 3))
  kl_spherical     model/spherical_vae.py:141-184 (three regions written through boolean masks)
  SphericalVAERef  model/spherical_vae.py:10-138 (softplus MLPs, kappa = softplus + 1, ELBO)
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def rsample(loc, scale, sample_shape=(), noise=None):
    """z of shape broadcast(sample_shape + scale.shape, sample_shape + loc.shape[:-1]) + (3,); noise = (u, eps) replays
    given draws, otherwise torch.rand then torch.randn as the reference (:56, :75)"""
    sample_shape = torch.Size(sample_shape)
    loc = loc / (loc.norm(dim=-1, keepdim=True) + 1e-5)                              # :30
    if noise is None:
        u = torch.rand(sample_shape + scale.shape, dtype=loc.dtype, device=loc.device)
        eps = torch.randn(sample_shape + loc.shape, dtype=loc.dtype, device=loc.device)
    else:
        u, eps = noise
    terms = torch.stack([u.log(), torch.log1p(-u) - 2 * scale], dim=0)                # :58-62
    w = 1 + torch.logsumexp(terms, dim=0) / scale                                     # :63
    v = eps - (eps * loc).sum(-1, keepdim=True) * loc                                 # :77-78
    v = v / (v.norm(dim=-1, keepdim=True) + 1e-8)                                     # :79
    w_e = w.unsqueeze(-1)                                                             # :82
    return w_e * loc + torch.sqrt(torch.clamp(1 - w_e ** 2, 1e-10, 1.0)) * v, w      # :83-84


def kl_spherical(kappa):
    """KL(vMF_3(kappa) || U(S^2)) per element, piecewise: 0 below 1e-3, sinh / cosh form up to 20, asymptote beyond"""
    k = torch.clamp(kappa, min=1e-8)                                                  # :150
    out = torch.zeros_like(k)                                                         # :157
    large = k > 20.0
    mid = ~(k < 1e-3) & ~large
    if mid.any():                                                                     # :161-169
        km = k[mid]
        sh = torch.sinh(km) + 1e-8
        out[mid] = km.log() - sh.log() + km * torch.cosh(km) / sh - 1.0
    if large.any():                                                                   # :172-182
        kl_ = k[large]
        out[large] = kl_.log() - (kl_ - math.log(2.0)) + kl_ * (1.0 + 2.0 * torch.exp(-2.0 * kl_)) - 1.0
    return out


class SphericalVAERef(nn.Module):
    """The Spherical VAE with the latent path written in torch ops (latent_dim is 3: the reference's sampler asserts it)"""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        ew, dw, dev = cfg["encoder_width"], cfg["decoder_width"], cfg["device"]
        self.encoder_function = nn.Linear(cfg["extrinsic_dim"], ew, device=dev)
        self.encoder_linear_layers = nn.ModuleList([nn.Linear(ew, ew, device=dev) for _ in range(cfg["encoder_depth"])])
        self.function_z_mu = nn.Linear(ew, cfg["latent_dim"], device=dev)
        self.function_z_logvar = nn.Linear(ew, 1, device=dev)
        self.decoder_function = nn.Linear(cfg["latent_dim"], dw, device=dev)
        self.decoder_linear_layers = nn.ModuleList([nn.Linear(dw, dw, device=dev) for _ in range(cfg["decoder_depth"])])
        self.function_x_mu = nn.Linear(dw, cfg["extrinsic_dim"], device=dev)

    def forward(self, x, noise=None):
        b = self.cfg["sftbeta"]
        h = F.softplus(self.encoder_function(x), beta=b)
        for layer in self.encoder_linear_layers:
            h = F.softplus(layer(h), beta=b)
        z_mu = self.function_z_mu(h)
        z_kappa = F.softplus(self.function_z_logvar(h)) + 1                           # :70
        z, _ = rsample(z_mu, z_kappa, noise=noise)                                    # :81-83
        h = F.softplus(self.decoder_function(z), beta=b)
        for layer in self.decoder_linear_layers:
            h = F.softplus(layer(h), beta=b)
        return z, self.function_x_mu(h), (z_mu, z_kappa)

    def elbo(self, x, x_mu, params):
        kl = kl_spherical(params[1]).mean()                                           # :134
        rec = torch.mean((x - x_mu) ** 2)                                             # :136
        return self.cfg["gamma"] * rec + self.cfg["beta"] * kl                        # :138
