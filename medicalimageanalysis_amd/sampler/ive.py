"""Drop-in for the reference's sampler/ive.py (and its copy CNN_experiment/ive.py): same names
(`IveFunction`, `Ive`, `ive`), same call signature and gradient, computed by HIP kernels on the tensor's device
instead of a device->host->scipy->device round trip (reference: sampler/ive.py:18,28,36,50-51).
"""
import torch
import torch.nn as nn

from .. import _lib


def _launch_unary(fn_name, v, z):
    dev = _lib.require_hip(z)
    zc = z.detach().contiguous()
    out = torch.empty_like(zc)
    with _lib.on_device(dev):
        rc = getattr(_lib.lib(), fn_name)(float(v), _lib.ptr(zc), _lib.ptr(out), zc.numel(), _lib.dtype_code(zc),
                                          _lib.stream_ptr(dev))
    _lib.check(rc, fn_name)
    return out.view(z.shape)


class IveFunction(torch.autograd.Function):
    """ive(v, z) = exp(-|z|) I_v(z); v is a python number shared by all elements (sampler/ive.py:12-29).
    Backward: dz = ive(v-1, z) - (v/z_safe + sign z) ive(v, z)  (sampler/ive.py:31-63)."""

    @staticmethod
    def forward(ctx, v, z):
        ctx.v = v
        ctx.save_for_backward(z)
        return _launch_unary("vmf_ive_fwd", v, z)

    @staticmethod
    def backward(ctx, grad_output):
        v = ctx.v
        (z,) = ctx.saved_tensors
        dev = _lib.require_hip(z, grad_output)
        zc = z.detach().contiguous()
        go = grad_output.to(zc.dtype).expand_as(zc).contiguous()
        gz = torch.empty_like(zc)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf_ive_bwd(float(v), _lib.ptr(zc), _lib.ptr(go), _lib.ptr(gz), zc.numel(),
                                        _lib.dtype_code(zc), _lib.stream_ptr(dev))
        _lib.check(rc, "vmf_ive_bwd")
        return None, gz.view(z.shape)


class Ive(nn.Module):
    """Module wrapper so you can write Ive(v)(z)  (sampler/ive.py:66-75)."""

    def __init__(self, v):
        super().__init__()
        self.v = v

    def forward(self, z):
        return IveFunction.apply(self.v, z)


ive = IveFunction.apply


def log_ive(v, z):
    """log ive(v, z) for z > 0, finite where ive itself underflows (no reference counterpart; not differentiable
    here -- the vMF log-normaliser's gradient goes through `bessel_ratio`)."""
    return _launch_unary("vmf_log_ive_fwd", v, z)


class _BesselRatio(torch.autograd.Function):
    """A(z) = I_v(z) / I_{v-1}(z) with dA/dz = 1 - (2v - 1) A / z - A^2"""

    @staticmethod
    def forward(ctx, v, z):
        a = _launch_unary("vmf_bessel_ratio_fwd", v, z.detach())
        ctx.save_for_backward(z.detach(), a)
        ctx.v = float(v)
        return a

    @staticmethod
    def backward(ctx, g):
        z, a = ctx.saved_tensors
        return None, g * (1 - (2 * ctx.v - 1) * a / z - a * a)


def bessel_ratio(v, z):
    """I_v(z) / I_{v-1}(z), v >= 1 (no reference counterpart); differentiable in z."""
    if z.requires_grad:
        return _BesselRatio.apply(v, z)
    return _launch_unary("vmf_bessel_ratio_fwd", v, z)
