"""KL(vMF || uniform) terms of the reference's models as HIP kernels with autograd.

variants (include/vmf_hip.h, enum vmf_kl_variant):
  spherical  model/spherical_vae.py:141-184      torus  model/torus_vae.py:182-194
  circular   CNN_experiment/circular_VAE.py:118-130   s1  regular_cnn/s1_vae.py:80-88
  stable_m3 / stable_m2: the same quantities without the sinh / I0 overflow (no reference counterpart)
"""
import torch

from . import _lib

VARIANTS = {"spherical": _lib.KL_SPHERICAL_PIECEWISE, "torus": _lib.KL_TORUS_RAW, "circular": _lib.KL_CIRCULAR_EPS,
            "s1": _lib.KL_S1_I0I1, "stable_m3": _lib.KL_STABLE_M3, "stable_m2": _lib.KL_STABLE_M2}


class _KLPerSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kappa, variant):
        dev = _lib.require_hip(kappa)
        kc = kappa.detach().contiguous()
        out = torch.empty_like(kc)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf_kl_fwd(_lib.ptr(kc), _lib.ptr(out), None, None, 1.0, kc.numel(), variant,
                                       _lib.dtype_code(kc), _lib.stream_ptr(dev))
        _lib.check(rc, "vmf_kl_fwd")
        ctx.save_for_backward(kc)
        ctx.variant = variant
        return out.view(kappa.shape)

    @staticmethod
    def backward(ctx, gout):
        (kc,) = ctx.saved_tensors
        dev = kc.device
        g = gout.to(kc.dtype).expand(kc.shape).contiguous()
        dk = torch.empty_like(kc)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf_kl_bwd(_lib.ptr(kc), _lib.ptr(g), 1, 1.0, _lib.ptr(dk), kc.numel(), ctx.variant,
                                       _lib.dtype_code(kc), _lib.stream_ptr(dev))
        _lib.check(rc, "vmf_kl_bwd")
        return dk, None


class _KLReduced(torch.autograd.Function):
    """alpha * sum_i KL(kappa_i) in one pass + a 1-block finish; the sum is carried in double."""

    @staticmethod
    def forward(ctx, kappa, variant, alpha):
        dev = _lib.require_hip(kappa)
        kc = kappa.detach().contiguous()
        ws = _lib.workspace(dev, _lib.KL_WORKSPACE_DOUBLES)
        acc = torch.empty(1, device=dev, dtype=torch.float64)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf_kl_fwd(_lib.ptr(kc), None, _lib.ptr(acc), _lib.ptr(ws), float(alpha), kc.numel(),
                                       variant, _lib.dtype_code(kc), _lib.stream_ptr(dev))
        _lib.check(rc, "vmf_kl_fwd")
        ctx.save_for_backward(kc)
        ctx.variant, ctx.alpha, ctx.shape = variant, float(alpha), kappa.shape
        return acc.to(kc.dtype).reshape(())

    @staticmethod
    def backward(ctx, gout):
        (kc,) = ctx.saved_tensors
        dev = kc.device
        g = gout.to(kc.dtype).reshape(1).contiguous()
        dk = torch.empty_like(kc)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf_kl_bwd(_lib.ptr(kc), _lib.ptr(g), 0, ctx.alpha, _lib.ptr(dk), kc.numel(),
                                       ctx.variant, _lib.dtype_code(kc), _lib.stream_ptr(dev))
        _lib.check(rc, "vmf_kl_bwd")
        return dk.view(ctx.shape), None, None


def kl_per_sample(kappa, variant="spherical"):
    """per-element KL with kappa's shape (what the models' kl_vmf_spherical_uniform returns)"""
    return _KLPerSample.apply(kappa, VARIANTS[variant])


def kl_mean(kappa, variant="spherical"):
    """kl_per_sample(kappa).mean() fused (model/spherical_vae.py:133-134)"""
    n = kappa.numel()
    return _KLReduced.apply(kappa, VARIANTS[variant], 1.0 / n if n else float("nan"))


def kl_sum(kappa, variant="s1"):
    """kl_per_sample(kappa).sum() fused (regular_cnn/s1_vae.py:90)"""
    return _KLReduced.apply(kappa, VARIANTS[variant], 1.0)
