"""VAE with a von Mises-Fisher latent on S^{latent_dim-1} for any latent_dim >= 2.

No reference counterpart: the reference's SphericalVAE can only run latent_dim == 3 (its sampler asserts m == 3,
sampler/von_mises_fisher.py:28).  Same encoder/decoder bodies and config keys as model/spherical_vae.py:17-47; the
posterior is VonMisesFisher(normalised z_mu, kappa) with kappa = softplus(.) + 1, one sample per row (z is (B, latent_dim),
not the (B, B, 3) broadcast), and the KL against the uniform prior comes from the general-m entropy kernel."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..sampler.spherical_uniform import HypersphericalUniform
from ..sampler.von_mises_fisher import VonMisesFisher, vmf_rsample_from_heads


class HypersphericalVAE(nn.Module):
    def __init__(self, config, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.config = config
        self.extrinsic_dim = config["extrinsic_dim"]
        self.latent_dim = config["latent_dim"]
        self.sftbeta = config["sftbeta"]
        self.device = config["device"]
        ew, dw = config["encoder_width"], config["decoder_width"]
        dev = self.device
        self.encoder_function = nn.Linear(self.extrinsic_dim, ew, device=dev)
        self.encoder_linear_layers = nn.ModuleList([nn.Linear(ew, ew, device=dev) for _ in range(config["encoder_depth"])])
        self.function_z_mu = nn.Linear(ew, self.latent_dim, device=dev)
        self.function_z_logvar = nn.Linear(ew, 1, device=dev)
        self.decoder_function = nn.Linear(self.latent_dim, dw, device=dev)
        self.decoder_linear_layers = nn.ModuleList([nn.Linear(dw, dw, device=dev) for _ in range(config["decoder_depth"])])
        self.function_x_mu = nn.Linear(dw, self.extrinsic_dim, device=dev)

    def encode(self, x):
        h = F.softplus(self.encoder_function(x), beta=self.sftbeta)
        for layer in self.encoder_linear_layers:
            h = F.softplus(layer(h), beta=self.sftbeta)
        z_mu = F.normalize(self.function_z_mu(h), dim=-1)
        z_kappa = F.softplus(self.function_z_logvar(h)).squeeze(-1) + 1
        return z_mu, z_kappa

    def reparemetrize(self, posteriori_params):
        z_mu, z_kappa = posteriori_params
        return VonMisesFisher(z_mu, z_kappa, validate_args=False).rsample()

    def decode(self, z):
        h = F.softplus(self.decoder_function(z), beta=self.sftbeta)
        for layer in self.decoder_linear_layers:
            h = F.softplus(layer(h), beta=self.sftbeta)
        return self.function_x_mu(h)

    def forward(self, x):
        """encode -> sample -> decode.  The sampler takes the raw outputs of BOTH heads: it normalises the mean rows and
        evaluates kappa = softplus(.) + 1 inside its kernels (forward and backward), so neither the (B, latent_dim)
        normalisation pass nor the softplus / add passes and their backward launches exist; the posterior parameters
        returned are the same (unit z_mu, kappa) that encode() gives."""
        h = F.softplus(self.encoder_function(x), beta=self.sftbeta)
        for layer in self.encoder_linear_layers:
            h = F.softplus(layer(h), beta=self.sftbeta)
        raw = self.function_z_mu(h)
        z, z_kappa = vmf_rsample_from_heads(raw, self.function_z_logvar(h).squeeze(-1))
        with torch.no_grad():  # reporting only: the KL term depends on kappa alone
            z_mu = F.normalize(raw, dim=-1)
        return z, self.decode(z), (z_mu, z_kappa)

    def _elbo(self, x, x_mu, posteriori_params):
        z_mu, z_kappa = posteriori_params
        q_z = VonMisesFisher(z_mu, z_kappa, validate_args=False)
        p_z = HypersphericalUniform(self.latent_dim - 1, device=x.device, dtype=x.dtype)
        KL = torch.distributions.kl_divergence(q_z, p_z).mean()
        reconstruction_loss = torch.mean((x - x_mu) * (x - x_mu))
        return self.config["gamma"] * reconstruction_loss + self.config["beta"] * KL
