"""Drop-in for the reference's CNN_experiment/circular_VAE.py (kappa squeezed to (B,), so z is (B, 3))."""
import torch
import torch.nn as nn

from ..sampler.von_mises_fisher import VonMisesFisher3DLogProb as VonMisesFisher3D
from ..sampler.von_mises_fisher import vmf3_rsample_from_heads
from .. import kl as _kl
from .. import losses as _losses


class CircularVAE(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.device = config["device"]
        self.extrinsic_dim = config["extrinsic_dim"]
        self.encoder_net = nn.Sequential(
            nn.Softplus(beta=1.0),
            nn.Linear(self.extrinsic_dim, config["encoder_width"]),
            nn.Softplus(beta=1.0),
            nn.Linear(config["encoder_width"], config["encoder_width"]),
            nn.Softplus(beta=1.0),
            nn.Linear(config["encoder_width"], config["encoder_width"]),
        )
        self.fc_mu = nn.Linear(config["encoder_width"], 3)
        self.fc_logvar = nn.Linear(config["encoder_width"], 1)
        self.decoder_net = nn.Sequential(
            nn.Linear(3, config["decoder_width"]),
            nn.Softplus(beta=1.0),
            nn.Linear(config["decoder_width"], config["decoder_width"]),
            nn.Softplus(beta=1.0),
            nn.Linear(config["decoder_width"], self.extrinsic_dim),
        )

    def encode(self, x):
        h = self.encoder_net(x)
        mu = self.fc_mu(h)
        logvar = nn.functional.softplus(self.fc_logvar(h)) + 1
        return mu, logvar.squeeze(-1)  # CNN_experiment/circular_VAE.py:47-53

    def reparameterize(self, mu, logvar):
        q_z = VonMisesFisher3D(mu, logvar)
        return q_z.rsample(), (mu, logvar)  # :74-76

    def decode(self, z):
        return self.decoder_net(z)

    def forward(self, x):
        """CNN_experiment/circular_VAE.py:81-88; kappa = softplus(.) + 1 is evaluated inside the sampler kernel"""
        h = self.encoder_net(x)
        mu = self.fc_mu(h)
        z_sample, logvar = vmf3_rsample_from_heads(mu, self.fc_logvar(h).squeeze(-1))
        x_recon = self.decode(z_sample)
        return z_sample, x_recon, (mu, logvar)

    def _elbo(self, x, x_recon, posterior_params):
        z_mu, z_kappa = posterior_params
        q_z = VonMisesFisher3D(z_mu, z_kappa)
        return _losses.model_elbo(self, CircularVAE, x, x_recon, q_z.scale, "circular")

    def kl_vmf_spherical_uniform(self, kappa) -> torch.Tensor:
        return _kl.kl_per_sample(kappa, "circular")  # :118-130
