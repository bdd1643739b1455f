"""The steps on either side of the latent sampler in the reference's train step, as fused HIP kernels with autograd:

  elbo_loss    gamma * mean((x - x_hat)^2) + beta * mean(KL(kappa))      model/spherical_vae.py:119-138,
               model/torus_vae.py:160-180, CNN_experiment/circular_VAE.py:90-116; with reduction="sum" the
               S^1-VAE's mse_sum + KLD (regular_cnn/s1_vae.py:70-92)
  build_torus  embedding of two sphere draws                               model/torus_vae.py:60-81
  latent_loss  supervised geodesic losses for the S1 / S2 / T2 datasets    dataloader/utils.py:101-180

There is no CPU path: tensors must live on a HIP device.
"""
import torch

from . import _lib
from .kl import VARIANTS


def _scalar_grad(gout, ref):
    return gout.detach().to(ref.dtype).reshape(1).contiguous()


class _Elbo(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, x_hat, kappa, a_rec, a_kl, variant):
        dev = _lib.require_hip(x, x_hat, kappa)
        if x.dtype != x_hat.dtype or kappa.dtype != x_hat.dtype:
            raise TypeError(f"elbo_loss: dtypes differ ({x.dtype}, {x_hat.dtype}, {kappa.dtype})")
        xs, hs = tuple(x.shape), tuple(x_hat.shape)
        tiled = len(hs) >= len(xs) and hs[len(hs) - len(xs):] == xs and x.numel() > 0
        if tiled:
            xc, hc = x.detach().contiguous(), x_hat.detach().contiguous()
        else:  # any other broadcast: materialise it (x - x_hat would do the same)
            xb, hb = torch.broadcast_tensors(x.detach(), x_hat.detach())
            xc, hc = xb.contiguous(), hb.contiguous()
        kc = kappa.detach().contiguous()
        loss = torch.empty((), device=dev, dtype=hc.dtype)
        ws = _lib.workspace(dev, _lib.ELBO_WORKSPACE_DOUBLES)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf_elbo_fwd(_lib.ptr(xc), _lib.ptr(hc), _lib.ptr(kc), _lib.ptr(loss), None, _lib.ptr(ws),
                                         a_rec, a_kl, xc.numel(), hc.numel(), kc.numel(), variant,
                                         _lib.dtype_code(hc), _lib.stream_ptr(dev))
        _lib.check(rc, "vmf_elbo_fwd")
        ctx.save_for_backward(xc, hc, kc)
        ctx.meta = (a_rec, a_kl, variant, xs, hs, tuple(kappa.shape), tiled)
        return loss

    @staticmethod
    def backward(ctx, gout):
        xc, hc, kc = ctx.saved_tensors
        a_rec, a_kl, variant, xs, hs, ks, tiled = ctx.meta
        dev = hc.device
        need_x, need_h, need_k = ctx.needs_input_grad[:3]
        g = _scalar_grad(gout, hc)
        dh = torch.empty_like(hc) if (need_h or need_x) else None
        dk = torch.empty_like(kc) if need_k else None
        with _lib.on_device(dev):
            rc = _lib.lib().vmf_elbo_bwd(_lib.ptr(xc), _lib.ptr(hc), _lib.ptr(kc), _lib.ptr(g), _lib.ptr(dh),
                                         _lib.ptr(dk), a_rec, a_kl, xc.numel(), hc.numel(), kc.numel(), variant,
                                         _lib.dtype_code(hc), _lib.stream_ptr(dev))
        _lib.check(rc, "vmf_elbo_bwd")
        dx = None
        if need_x:  # data rarely needs a gradient: d/dx = -sum of d/dx_hat over the broadcast dimensions
            dx = (-dh).sum_to_size(xs) if dh.shape != torch.Size(xs) else -dh
        if dh is not None and not need_h:
            dh = None
        if dh is not None and dh.shape != torch.Size(hs):
            dh = dh.sum_to_size(hs)
        return dx, dh, (dk.view(ks) if dk is not None else None), None, None, None


def elbo_loss(x, x_hat, kappa, gamma=1.0, beta=1.0, variant="spherical", reduction="mean"):
    """gamma * mean((x - x_hat)^2) + beta * mean(KL_variant(kappa)) in one fused reduction; `reduction="sum"` uses
    sums for both terms.  The mean of the squared error runs over the BROADCAST shape of x - x_hat."""
    n_hat = torch.broadcast_shapes(x.shape, x_hat.shape).numel()
    n_k = kappa.numel()
    if reduction == "mean":
        a_rec = float(gamma) / n_hat if n_hat else float("nan")
        a_kl = float(beta) / n_k if n_k else float("nan")
    elif reduction == "sum":
        a_rec, a_kl = float(gamma), float(beta)
    else:
        raise ValueError(f"reduction must be 'mean' or 'sum', got {reduction!r}")
    return _Elbo.apply(x, x_hat, kappa, a_rec, a_kl, VARIANTS[variant])


def model_elbo(model, owner, x, x_hat, kappa, variant):
    """`_elbo` of the VAE classes: gamma * mean((x - x_hat)^2) + beta * mean(model.kl_vmf_spherical_uniform(kappa))
    (model/spherical_vae.py:132-138).  While the KL method is the one `owner` (the class of this package) defines, both
    terms are one fused reduction with that method's formula; a subclass that overrides the method gets exactly the
    reference's composition -- its own KL, `.mean()`, and the squared error over the broadcast shape -- so the override
    is honoured as it would be in the reference."""
    gamma, beta = model.config["gamma"], model.config["beta"]
    if type(model).kl_vmf_spherical_uniform is owner.kl_vmf_spherical_uniform:
        return elbo_loss(x, x_hat, kappa, gamma, beta, variant)
    kl = model.kl_vmf_spherical_uniform(kappa).mean()
    recon = torch.mean((x - x_hat) ** 2)
    return gamma * recon + beta * kl


class _BuildTorus(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z_theta, z_phi, R, r):
        dev = _lib.require_hip(z_theta, z_phi)
        if z_theta.dtype != z_phi.dtype:
            raise TypeError("build_torus: dtypes differ")
        ts, ps = tuple(z_theta.shape), tuple(z_phi.shape)
        if ts != ps:
            if ts[-1] != ps[-1]:
                raise ValueError(f"build_torus: last dimensions differ ({ts[-1]} vs {ps[-1]})")
            tb, pb = torch.broadcast_tensors(z_theta.detach(), z_phi.detach())
        else:
            tb, pb = z_theta.detach(), z_phi.detach()
        if tb.shape[-1] < 2:
            raise IndexError("build_torus: the last dimension must hold at least (cos, sin)")
        tc, pc = tb.contiguous(), pb.contiguous()
        zdim = tc.shape[-1]
        n = tc.numel() // zdim
        out = torch.empty(tc.shape[:-1] + (3,), device=dev, dtype=tc.dtype)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf_build_torus_fwd(_lib.ptr(tc), _lib.ptr(pc), _lib.ptr(out), n, zdim, R, r,
                                                _lib.dtype_code(tc), _lib.stream_ptr(dev))
        _lib.check(rc, "vmf_build_torus_fwd")
        ctx.save_for_backward(tc, pc)
        ctx.meta = (R, r, ts, ps)
        return out

    @staticmethod
    def backward(ctx, gout):
        tc, pc = ctx.saved_tensors
        R, r, ts, ps = ctx.meta
        dev = tc.device
        g = gout.to(tc.dtype).contiguous()
        gt, gp = torch.empty_like(tc), torch.empty_like(pc)
        zdim = tc.shape[-1]
        with _lib.on_device(dev):
            rc = _lib.lib().vmf_build_torus_bwd(_lib.ptr(tc), _lib.ptr(pc), _lib.ptr(g), _lib.ptr(gt), _lib.ptr(gp),
                                                tc.numel() // zdim, zdim, R, r, _lib.dtype_code(tc),
                                                _lib.stream_ptr(dev))
        _lib.check(rc, "vmf_build_torus_bwd")
        if tuple(gt.shape) != ts:
            gt = gt.sum_to_size(ts)
        if tuple(gp.shape) != ps:
            gp = gp.sum_to_size(ps)
        return gt, gp, None, None


def build_torus(z_theta, z_phi, longest_radius, shortest_radius):
    """((R + r cos t) cos p, (r + r cos t) sin p, r sin t) with (cos, sin) = the first two components of each
    input (model/torus_vae.py:60-81; y keeps the reference's `r + r cos t`)."""
    return _BuildTorus.apply(z_theta, z_phi, float(longest_radius), float(shortest_radius))


_LATENT = {"S1_dataset": _lib.LATENT_S1, "S2_dataset": _lib.LATENT_S2, "T2_dataset": _lib.LATENT_T2}


class _LatentLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, labels, variant, R, r):
        dev = _lib.require_hip(z, labels)
        zc = z.detach().contiguous()
        lc = labels.detach().to(zc.dtype).contiguous()
        zdim = zc.shape[-1]
        n = zc.numel() // zdim
        out = torch.empty((), device=dev, dtype=zc.dtype)
        ws = _lib.workspace(dev, _lib.KL_WORKSPACE_DOUBLES)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf_latent_loss_fwd(_lib.ptr(zc), _lib.ptr(lc), None, _lib.ptr(out), _lib.ptr(ws), n, zdim,
                                                lc.shape[0], lc.shape[1], variant, R, r, _lib.dtype_code(zc),
                                                _lib.stream_ptr(dev))
        _lib.check(rc, "vmf_latent_loss_fwd")
        ctx.save_for_backward(zc, lc)
        ctx.meta = (variant, R, r, tuple(z.shape))
        return out

    @staticmethod
    def backward(ctx, gout):
        zc, lc = ctx.saved_tensors
        variant, R, r, zs = ctx.meta
        dev = zc.device
        g = _scalar_grad(gout, zc)
        dz = torch.empty_like(zc)
        zdim = zc.shape[-1]
        with _lib.on_device(dev):
            rc = _lib.lib().vmf_latent_loss_bwd(_lib.ptr(zc), _lib.ptr(lc), _lib.ptr(g), _lib.ptr(dz),
                                                zc.numel() // zdim, zdim, lc.shape[0], lc.shape[1], variant, R, r,
                                                _lib.dtype_code(zc), _lib.stream_ptr(dev))
        _lib.check(rc, "vmf_latent_loss_bwd")
        return dz.view(zs), None, None, None, None


def latent_loss(labels, z, config):
    """Same signature and values as dataloader/utils.py:101-180: mean geodesic loss of the latent code against the
    ground-truth angles in `labels` (B, L).  z is (..., B, C): leading dimensions broadcast against the labels the
    way the reference's `(B, B, 3)` sample does."""
    dataset = config["dataset"]
    if dataset not in _LATENT:
        raise ValueError("Unknown dataset type in latent_loss()")
    if labels.dim() != 2:
        raise IndexError(f"latent_loss: labels must be (B, L), got {tuple(labels.shape)}")
    if z.dim() < 2 or z.shape[-2] != labels.shape[0]:
        # the reference's subtraction broadcasts z[..., c] (…, B) against labels[:, k] (B,)
        if not (z.dim() >= 2 and (z.shape[-2] == 1 or labels.shape[0] == 1)):
            raise RuntimeError(f"latent_loss: z {tuple(z.shape)} does not broadcast against {labels.shape[0]} labels")
        z = z.expand(z.shape[:-2] + (max(z.shape[-2], labels.shape[0]), z.shape[-1]))
        labels = labels.expand(z.shape[-2], labels.shape[1])
    R = float(config.get("longest_radius", 0.0)) if dataset == "T2_dataset" else 0.0
    r = float(config.get("shortest_radius", 0.0)) if dataset == "T2_dataset" else 0.0
    if dataset == "T2_dataset":
        R, r = float(config["longest_radius"]), float(config["shortest_radius"])
    return _LatentLoss.apply(z, labels, _LATENT[dataset], R, r)
