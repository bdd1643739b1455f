"""Drop-in for the reference's model/torus_vae.py (same names, including the attribute spellings
`shortes_radius` and `fucntion_z_phi_logvar`).  geomstats is not needed: `gs.stack` is `torch.stack`."""
import torch
import torch.nn as nn
import torch.nn.functional as F
import numpy as np  # noqa: F401

from ..sampler.spherical_uniform import *  # noqa: F401,F403
from ..sampler.von_mises_fisher import *  # noqa: F401,F403
from ..sampler.spherical_uniform import SphericalUniform
from ..sampler.von_mises_fisher import VonMisesFisher3D, vmf3_rsample_from_heads
from .. import kl as _kl
from .. import losses as _losses


class ToroidalVAE(nn.Module):
    def __init__(self, config, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
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
        self.longest_radius = config["longest_radius"]
        self.shortes_radius = config["shortest_radius"]

        dev = self.device
        self.encoder_function = nn.Linear(self.extrinsic_dim, self.encoder_width, device=dev)
        self.encoder_linear_layers = nn.ModuleList(
            [nn.Linear(self.encoder_width, self.encoder_width, device=dev) for _ in range(self.encoder_depth)])
        self.function_z_theta_mu = nn.Linear(self.encoder_width, self.latent_dim, device=dev)
        self.function_z_theta_logvar = nn.Linear(self.encoder_width, 1, device=dev)
        self.function_z_phi_mu = nn.Linear(self.encoder_width, self.latent_dim, device=dev)
        self.fucntion_z_phi_logvar = nn.Linear(self.encoder_width, 1, device=dev)
        self.decoder_function = nn.Linear(self.latent_dim, self.decoder_width, device=dev)
        self.decoder_linear_layers = nn.ModuleList(
            [nn.Linear(self.decoder_width, self.decoder_width, device=dev) for _ in range(self.decoder_depth)])
        self.function_x_mu = nn.Linear(self.decoder_width, self.extrinsic_dim, device=dev)
        self.dropout = nn.Dropout(p=self.dropout_p)

    def _build_torus(self, z_theta, z_phi):
        """model/torus_vae.py:60-81: the first two components of each sphere sample act as (cos, sin);
        y uses (r + r cos theta), as the reference does."""
        return _losses.build_torus(z_theta, z_phi, self.longest_radius, self.shortes_radius)

    def encode(self, x):
        h = F.softplus(self.encoder_function(x), beta=self.sftbeta)
        for layer in self.encoder_linear_layers:
            h = F.softplus(layer(h), beta=self.sftbeta)
        z_theta_mu = self.function_z_theta_mu(h)
        z_theta_logvar = F.softplus(self.function_z_theta_logvar(h)) + 1
        z_phi_mu = self.function_z_phi_mu(h)
        z_phi_logvar = F.softplus(self.fucntion_z_phi_logvar(h)) + 1
        return z_theta_mu, z_theta_logvar, z_phi_mu, z_phi_logvar

    def reparemetrize(self, posteriori_params):
        """two independent draws, theta first then phi (model/torus_vae.py:109-124)"""
        z_theta_mu, z_theta_kappa, z_phi_mu, z_phi_kappa = posteriori_params
        q_theta_z = VonMisesFisher3D(z_theta_mu, z_theta_kappa)
        q_phi_z = VonMisesFisher3D(z_phi_mu, z_phi_kappa)
        z_theta = q_theta_z.rsample()
        z_phi = q_phi_z.rsample()
        return self._build_torus(z_theta, z_phi)

    def decode(self, z):
        h = F.softplus(self.decoder_function(z), beta=self.sftbeta)
        for layer in self.decoder_linear_layers:
            h = F.softplus(layer(h), beta=self.sftbeta)
        return self.function_x_mu(h)

    def forward(self, x):
        """encode -> reparemetrize -> decode (model/torus_vae.py:144-158); kappa = softplus(.) + 1 of both heads is
        evaluated inside the sampler kernels (see SphericalVAE.forward).  Draw order: theta, then phi."""
        h = F.softplus(self.encoder_function(x), beta=self.sftbeta)
        for layer in self.encoder_linear_layers:
            h = F.softplus(layer(h), beta=self.sftbeta)
        z_theta_mu, z_phi_mu = self.function_z_theta_mu(h), self.function_z_phi_mu(h)
        z_theta, z_theta_kappa = vmf3_rsample_from_heads(z_theta_mu, self.function_z_theta_logvar(h))
        z_phi, z_phi_kappa = vmf3_rsample_from_heads(z_phi_mu, self.fucntion_z_phi_logvar(h))
        z = self._build_torus(z_theta, z_phi)
        return z, self.decode(z), (z_theta_mu, z_theta_kappa, z_phi_mu, z_phi_kappa)

    def _elbo(self, x, x_mu, posteriori_params):
        """called once per angle pair by the training loop (model/train.py:74-77)"""
        z_mu, z_kappa = posteriori_params
        q_z = VonMisesFisher3D(z_mu, z_kappa)
        p_z = SphericalUniform(dim=self.latent_dim - 1, device=x.device)  # noqa: F841
        return _losses.model_elbo(self, ToroidalVAE, x, x_mu, q_z.scale, "torus")

    def kl_vmf_spherical_uniform(self, kappa) -> torch.Tensor:
        """unguarded form of model/torus_vae.py:182-194 (NaN once sinh overflows, like the reference)"""
        return _kl.kl_per_sample(kappa, "torus")
