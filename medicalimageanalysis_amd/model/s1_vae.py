"""Drop-in for the reference's regular_cnn/s1_vae.py (identical copy: invariant_cnn/s1_vae.py): `S1_VAE` with the
projected-normal reparameterisation and `loss_function` with the von Mises KL."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, rng
from .. import kl as _kl
from .. import losses as _losses


class _S1ProjNormal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean_vec, kappa, noise, seed, offset, offset_dev=None):
        dev = _lib.require_hip(mean_vec, kappa, noise)
        n, dim = mean_vec.shape
        mc = mean_vec.detach().contiguous()
        kc = kappa.detach().to(mc.dtype).reshape(-1).contiguous()
        nc = None if noise is None else noise.contiguous()
        z = torch.empty_like(mc)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf_s1_projnormal_fwd(_lib.ptr(mc), _lib.ptr(kc), _lib.ptr(nc), _lib.ptr(z), n, dim, 0,
                                                  seed, offset, _lib.ptr(offset_dev), _lib.dtype_code(mc),
                                                  _lib.stream_ptr(dev))
        _lib.check(rc, "vmf_s1_projnormal_fwd")
        ctx.save_for_backward(mc, kc, nc)
        ctx.meta = (seed, offset, kappa.shape)
        ctx.offset_dev = offset_dev
        return z

    @staticmethod
    def backward(ctx, gz):
        mc, kc, nc = ctx.saved_tensors
        seed, offset, kshape = ctx.meta
        n, dim = mc.shape
        dev = mc.device
        g = gz.to(mc.dtype).contiguous()
        dm, dk = torch.empty_like(mc), torch.empty_like(kc)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf_s1_projnormal_bwd(_lib.ptr(mc), _lib.ptr(kc), _lib.ptr(nc), _lib.ptr(g), _lib.ptr(dm),
                                                  _lib.ptr(dk), n, dim, 0, seed, offset, _lib.ptr(ctx.offset_dev),
                                                  _lib.dtype_code(mc),
                                                  _lib.stream_ptr(dev))
        _lib.check(rc, "vmf_s1_projnormal_bwd")
        return dm, dk.view(kshape), None, None, None, None


def s1_reparameterize(mean_vec, kappa, noise=None):
    """z = F.normalize(mean_vec * kappa + noise, p=2, dim=1)  (regular_cnn/s1_vae.py:48-55)"""
    if noise is None and rng.rng_mode() == "torch":
        noise = torch.randn_like(mean_vec)  # the reference's draw (:51)
    seed, offset, offset_dev = (0, 0, None) if noise is not None else rng.default_generator.ticket(mean_vec.device)
    return _S1ProjNormal.apply(mean_vec, kappa, noise, seed, offset, offset_dev)


class S1_VAE(nn.Module):
    def __init__(self, input_dim, hidden_dim=64):
        super(S1_VAE, self).__init__()
        self.enc_fc1 = nn.Linear(input_dim, hidden_dim)
        self.enc_fc2 = nn.Linear(hidden_dim, hidden_dim)
        self.z_mean = nn.Linear(hidden_dim, 2)
        self.z_kappa = nn.Linear(hidden_dim, 1)
        self.dec_fc1 = nn.Linear(2, hidden_dim)
        self.dec_fc2 = nn.Linear(hidden_dim, hidden_dim)
        self.dec_out = nn.Linear(hidden_dim, input_dim)
        self.act = nn.Softplus()

    def encode(self, x):
        h = self.act(self.enc_fc1(x))
        h = self.act(self.enc_fc2(h))
        mean_vec = F.normalize(self.z_mean(h), p=2, dim=1)
        kappa = F.softplus(self.z_kappa(h)) + 1.0
        return mean_vec, kappa

    def reparameterize(self, mean_vec, kappa):
        return s1_reparameterize(mean_vec, kappa)

    def decode(self, z):
        h = self.act(self.dec_fc1(z))
        h = self.act(self.dec_fc2(h))
        return self.dec_out(h)

    def forward(self, x):
        mean_vec, kappa = self.encode(x)
        z = self.reparameterize(mean_vec, kappa)
        x_recon = self.decode(z)
        return x_recon, mean_vec, kappa, z


def loss_function(recon_x, x, mean_vec, kappa):
    """sum-MSE + sum_i (k I1(k)/I0(k) - log I0(k))  (regular_cnn/s1_vae.py:70-92); overflows to NaN where the
    reference does (k > ~88.7 in fp32)."""
    if recon_x.shape != x.shape:  # F.mse_loss broadcasts (with a warning); keep that path on stock ops
        return F.mse_loss(recon_x, x, reduction="sum") + _kl.kl_sum(kappa.squeeze(), "s1")
    return _losses.elbo_loss(x, recon_x, kappa.squeeze(), 1.0, 1.0, "s1", reduction="sum")
