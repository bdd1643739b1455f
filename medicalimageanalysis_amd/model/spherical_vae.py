"""Drop-in for the reference's model/spherical_vae.py: same class, config keys, method names (including the
`reparemetrize` spelling) and return values.  The MLP bodies stay stock PyTorch (dense GEMMs on hipBLASLt); the
latent sampler and the KL term run in libvmf_hip.so."""
import torch
import torch.nn as nn
import torch.nn.functional as F
import numpy as np  # noqa: F401  (the reference module exposes it)

from ..sampler.spherical_uniform import *  # noqa: F401,F403
from ..sampler.von_mises_fisher import *  # noqa: F401,F403
from ..sampler.spherical_uniform import SphericalUniform
from ..sampler.von_mises_fisher import VonMisesFisher3D, vmf3_rsample_from_heads
from .. import kl as _kl
from .. import losses as _losses


class SphericalVAE(nn.Module):
    def __init__(self, config, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        # same keys as model/spherical_vae.py:17-28
        self.config = config
        self.extrinsic_dim = config["extrinsic_dim"]
        self.latent_dim = config["latent_dim"]
        self.sftbeta = config["sftbeta"]
        self.encoder_width = config["encoder_width"]
        self.encoder_depth = config["encoder_depth"]
        self.decoder_width = config["decoder_width"]
        self.decoder_depth = config["decoder_depth"]
        self.dropout_p = config["dropout_p"]
        self.device = config["device"]

        dev = self.device
        self.encoder_function = nn.Linear(self.extrinsic_dim, self.encoder_width, device=dev)
        self.encoder_linear_layers = nn.ModuleList(
            [nn.Linear(self.encoder_width, self.encoder_width, device=dev) for _ in range(self.encoder_depth)])
        self.function_z_mu = nn.Linear(self.encoder_width, self.latent_dim, device=dev)
        self.function_z_logvar = nn.Linear(self.encoder_width, 1, device=dev)
        self.decoder_function = nn.Linear(self.latent_dim, self.decoder_width, device=dev)
        self.decoder_linear_layers = nn.ModuleList(
            [nn.Linear(self.decoder_width, self.decoder_width, device=dev) for _ in range(self.decoder_depth)])
        self.function_x_mu = nn.Linear(self.decoder_width, self.extrinsic_dim, device=dev)
        self.dropout = nn.Dropout(p=self.dropout_p)

    def encode(self, x):
        """(z_mu, z_kappa): un-normalised mean direction (B, latent_dim) and kappa = softplus(.) + 1 of shape (B, 1)
        (model/spherical_vae.py:51-72)"""
        h = F.softplus(self.encoder_function(x), beta=self.sftbeta)
        for layer in self.encoder_linear_layers:
            h = F.softplus(layer(h), beta=self.sftbeta)
        z_mu = self.function_z_mu(h)
        z_kappa = F.softplus(self.function_z_logvar(h)) + 1
        return z_mu, z_kappa

    def reparemetrize(self, posteriori_params):
        """model/spherical_vae.py:74-83.  kappa is (B, 1), so the sample is (B, B, 3) as in the reference."""
        z_mu, z_kappa = posteriori_params
        q_z = VonMisesFisher3D(z_mu, z_kappa)
        return q_z.rsample()

    def decode(self, z):
        h = F.softplus(self.decoder_function(z), beta=self.sftbeta)
        for layer in self.decoder_linear_layers:
            h = F.softplus(layer(h), beta=self.sftbeta)
        return self.function_x_mu(h)

    def forward(self, x):
        """encode -> reparemetrize -> decode with the values and the noise order of model/spherical_vae.py:103-117.  The
        sampler kernel takes the RAW output of the concentration head and evaluates kappa = softplus(.) + 1 itself
        (handing kappa back for the KL term), so the head's softplus / add passes and their backward launches are
        gone; encode() and reparemetrize() remain for callers that use them separately."""
        h = F.softplus(self.encoder_function(x), beta=self.sftbeta)
        for layer in self.encoder_linear_layers:
            h = F.softplus(layer(h), beta=self.sftbeta)
        z_mu = self.function_z_mu(h)
        z, z_kappa = vmf3_rsample_from_heads(z_mu, self.function_z_logvar(h))
        x_mu = self.decode(z)
        return z, x_mu, (z_mu, z_kappa)

    def _elbo(self, x, x_mu, posteriori_params):
        """gamma * mean((x - x_mu)^2) + beta * mean(KL)  (model/spherical_vae.py:119-138).  The distribution
        objects are still constructed (argument checks, the m == 3 assertion); the squared error over the
        broadcast (B, B, D) shape and the KL mean are ONE fused reduction (no mask gathers, no host syncs)."""
        z_mu, z_kappa = posteriori_params
        q_z = VonMisesFisher3D(z_mu, z_kappa)
        p_z = SphericalUniform(dim=self.latent_dim - 1, device=x.device)  # noqa: F841  (built, unused: :130)
        return _losses.model_elbo(self, SphericalVAE, x, x_mu, q_z.scale, "spherical")

    def kl_vmf_spherical_uniform(self, kappa) -> torch.Tensor:
        """KL(vMF(mu, kappa) || Uniform(S^2)) per element, piecewise as model/spherical_vae.py:141-184."""
        return _kl.kl_per_sample(kappa, "spherical")
