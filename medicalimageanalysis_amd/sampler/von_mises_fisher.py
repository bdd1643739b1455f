"""Drop-in for the reference's sampler/von_mises_fisher.py (`VonMisesFisher3D`), the variant with `log_prob`
from CNN_experiment/von_mises_fisher.py (`VonMisesFisher3DLogProb`), and the general-m `VonMisesFisher`
(Wood rejection sampler + Householder reflection) the reference does not have.

Everything numerical runs in libvmf_hip.so on the tensors' HIP device; this file is shape logic, argument
checks and autograd glue.
"""
import math  # noqa: F401  (re-exported: the reference's model files reach `math` through `from sampler.von_mises_fisher import *`)

import functools
import os

import torch
from torch.distributions import Distribution, constraints
from torch.distributions.kl import register_kl

from .. import _lib, rng
from .ive import ive
from .spherical_uniform import HypersphericalUniform

__all__ = ["math", "torch", "ive", "Distribution", "constraints", "VonMisesFisher3D", "VonMisesFisher3DLogProb",
           "VonMisesFisher", "HypersphericalUniform", "vmf3_rsample_from_heads", "vmf_rsample_from_heads"]

_MAX_DIMS = 8


# ---- broadcast plan of VonMisesFisher3D.rsample --------------------------------------------------------
def _bstrides(shape, target):
    """element strides of a dense tensor of `shape` viewed as broadcast to `target` (0 on broadcast dims)"""
    nd = len(target)
    shape = (1,) * (nd - len(shape)) + tuple(shape)
    st, acc = [0] * nd, 1
    for d in range(nd - 1, -1, -1):
        if shape[d] != 1:
            assert shape[d] == target[d]
            st[d] = acc
        acc *= shape[d]
    return st


def vmf3_broadcast_plan(loc_shape, scale_shape, sample_shape=()):
    """Shapes/strides of `w.unsqueeze(-1) * loc + factor * v` (sampler/von_mises_fisher.py:82-84).

    w has shape sample_shape + scale.shape, loc/eps/v have shape sample_shape + loc.shape; the result is their
    broadcast -- (B, 3) for scale (B,), and the (B, B, 3) outer product for the (B, 1) scale the VAEs pass.
    Plans are cached per shape triple (a train loop asks for the same one every step); treat the dict as read-only."""
    return _plan_cached(tuple(loc_shape), tuple(scale_shape), tuple(sample_shape))


@functools.lru_cache(maxsize=512)
def _plan_cached(loc_shape, scale_shape, sample_shape):
    lrows = loc_shape[:-1]
    w_shape = sample_shape + scale_shape
    e_shape = sample_shape + lrows
    o_shape = tuple(torch.broadcast_shapes(w_shape, e_shape))  # raises RuntimeError like the reference's `*`
    odims = o_shape if len(o_shape) else (1,)
    if len(odims) > _MAX_DIMS:
        raise RuntimeError(f"rsample result has {len(odims)} batch dims; at most {_MAX_DIMS} are supported")
    pad = lambda s: s if len(o_shape) else [0]  # noqa: E731
    plan = {
        "o_shape": o_shape, "w_shape": w_shape, "e_shape": e_shape, "odims": list(odims),
        "w_str": pad(_bstrides(w_shape, o_shape)), "k_str": pad(_bstrides(scale_shape, o_shape)),
        "l_str": pad(_bstrides(lrows, o_shape)), "e_str": pad(_bstrides(e_shape, o_shape)),
    }
    # the ctypes arrays the entry points take, built once
    plan["c_args"] = (len(plan["odims"]),) + tuple(_lib.i64_array(plan[k]) for k in ("odims", "w_str", "k_str", "l_str", "e_str"))
    return plan


def _numel(shape):
    n = 1
    for s in shape:
        n *= int(s)
    return n


def _capture_safe(validate_args, t):
    """Distribution's argument check (`scale > 0`) reads its result on the host; a stream that is being captured
    into a HIP graph cannot synchronise, so the default check is skipped there (an explicit True is kept)."""
    if validate_args is None and t.is_cuda and torch.cuda.is_current_stream_capturing():
        return False
    return validate_args


def _plan_kind(plan):
    """(kind, n_i, n_j) of a broadcast plan as the library classifies it (vmf3_broadcast_kind; host only)"""
    if "kind" not in plan:
        ni, nj = _lib.C.c_int64(0), _lib.C.c_int64(0)
        kind = _lib.lib().vmf3_broadcast_kind(*plan["c_args"], _lib.C.byref(ni), _lib.C.byref(nj))
        if kind < 0:
            _lib.check(kind, "vmf3_broadcast_kind")
        plan["kind"] = (kind, ni.value, nj.value)
    return plan["kind"]


class _VMF3RSample(torch.autograd.Function):
    """z = rsample(loc, scale) for given or in-kernel noise.  With softplus=True `scale` is the RAW output of the
    concentration head, the kernels use kappa = softplus(raw) + 1, and the function returns (z, kappa): both outputs
    are differentiable and one backward launch serves both (no separate softplus / add passes in either direction)."""

    @staticmethod
    def forward(ctx, loc, scale, u, eps, plan, seed, offset, offset_dev=None, softplus=False, row0=0):
        dev = _lib.require_hip(loc, scale, u, eps)
        locc, scalec = loc.detach().contiguous(), scale.detach().contiguous()
        uc = None if u is None else u.contiguous()
        epsc = None if eps is None else eps.contiguous()
        z = torch.empty(plan["o_shape"] + (3,), device=dev, dtype=loc.dtype)
        kappa = torch.empty_like(scalec) if softplus else None
        flags = _lib.FLAG_SOFTPLUS_KAPPA if softplus else 0
        if z.numel():
            with _lib.on_device(dev):
                rc = _lib.lib().vmf3_rsample_fwd(
                    _lib.ptr(locc), _lib.ptr(scalec), _lib.ptr(uc), _lib.ptr(epsc), _lib.ptr(z), None, _lib.ptr(kappa),
                    *plan["c_args"], flags, _lib.dtype_code(loc), row0, seed, offset, _lib.ptr(offset_dev),
                    _lib.stream_ptr(dev))
            _lib.check(rc, "vmf3_rsample_fwd")
        elif softplus and kappa.numel():
            kappa = torch.nn.functional.softplus(scalec) + 1  # nothing to sample: no kernel ran
        ctx.save_for_backward(locc, scalec, uc, epsc)
        ctx.plan, ctx.seed, ctx.offset, ctx.offset_dev, ctx.flags, ctx.row0 = plan, seed, offset, offset_dev, flags, row0
        ctx.loc_shape, ctx.scale_shape = loc.shape, scale.shape
        return (z, kappa.view(scale.shape)) if softplus else z

    @staticmethod
    def backward(ctx, gz, gkappa=None):
        locc, scalec, uc, epsc = ctx.saved_tensors
        plan, flags = ctx.plan, ctx.flags
        dev = locc.device
        lib = _lib.lib()
        gzc = gz.to(locc.dtype).contiguous()
        gkc = None if gkappa is None else gkappa.to(locc.dtype).contiguous()
        odims, n_out = plan["odims"], _numel(plan["odims"])
        n_loc, n_scale = locc.numel() // 3, scalec.numel()
        kind, n_i, n_j = _plan_kind(plan) if n_out else (_lib.KIND_GENERAL, 0, 0)
        common = (flags, _lib.dtype_code(locc), ctx.row0, ctx.seed, ctx.offset, _lib.ptr(ctx.offset_dev),
                  _lib.stream_ptr(dev))
        if kind == _lib.KIND_OUTER:
            # the (B, 1)-scale broadcast: one pass over grad z, sums over i and over j inside the library (fixed order)
            dloc, dscale = torch.empty_like(locc), torch.empty_like(scalec)
            nbytes = _lib.C.c_int64(0)
            _lib.check(lib.vmf3_rsample_bwd_outer_workspace(n_i, n_j, _lib.dtype_code(locc), _lib.C.byref(nbytes)),
                       "vmf3_rsample_bwd_outer_workspace")
            nd = (nbytes.value + 7) // 8
            # batch-sized shapes share a cached scratch (no allocation inside a captured step); large ones come and go
            ws = _lib.workspace(dev, nd) if nd <= (1 << 17) else torch.empty(nd, device=dev, dtype=torch.float64)
            with _lib.on_device(dev):
                rc = lib.vmf3_rsample_bwd_outer(_lib.ptr(locc), _lib.ptr(scalec), _lib.ptr(uc), _lib.ptr(epsc),
                                                _lib.ptr(gzc), _lib.ptr(gkc), _lib.ptr(dloc), _lib.ptr(dscale), n_i, n_j,
                                                _lib.ptr(ws), ws.numel() * 8, *common)
            _lib.check(rc, "vmf3_rsample_bwd_outer")
            return (dloc.reshape(ctx.loc_shape), dscale.reshape(ctx.scale_shape)) + (None,) * 8
        shared = not (n_loc == n_out and n_scale == n_out)
        # Any other broadcast with rows that share a loc row / scale element (sample shapes ...): by default the kernel
        # writes one gradient row per output row and torch sums them (same bits every run); VMF_ATOMIC_BWD=1 lets the
        # kernel combine them with float atomics instead (no temporary, order varies run to run).
        per_row = shared and n_out > 0 and os.environ.get("VMF_ATOMIC_BWD", "0") != "1"
        if per_row:
            dloc = torch.empty(tuple(odims) + (3,), device=dev, dtype=locc.dtype)
            dscale = torch.empty(tuple(odims), device=dev, dtype=locc.dtype)
        else:
            dloc, dscale = torch.empty_like(locc), torch.empty_like(scalec)
        in_kernel_gk = gkc is not None and not shared and kind == _lib.KIND_ROWS
        if dloc.numel() or dscale.numel():
            with _lib.on_device(dev):
                rc = lib.vmf3_rsample_bwd(
                    _lib.ptr(locc), _lib.ptr(scalec), _lib.ptr(uc), _lib.ptr(epsc), _lib.ptr(gzc),
                    _lib.ptr(gkc) if in_kernel_gk else None, _lib.ptr(dloc), _lib.ptr(dscale),
                    n_out if per_row else n_loc, n_out if per_row else n_scale, *plan["c_args"], *common)
            _lib.check(rc, "vmf3_rsample_bwd")
        if per_row:
            ldims = [d for d, (n, st) in enumerate(zip(odims, plan["l_str"])) if st == 0 and n > 1]
            kdims = [d for d, (n, st) in enumerate(zip(odims, plan["k_str"])) if st == 0 and n > 1]
            dloc = dloc.sum(dim=ldims) if ldims else dloc
            dscale = dscale.sum(dim=kdims) if kdims else dscale
        dscale = dscale.reshape(ctx.scale_shape)
        if gkc is not None and not in_kernel_gk:  # rare shapes: the kappa output's own chain rule with torch ops
            dscale = dscale + gkc.reshape(ctx.scale_shape) * torch.sigmoid(scalec.reshape(ctx.scale_shape))
        return (dloc.reshape(ctx.loc_shape), dscale) + (None,) * 8


def vmf3_rsample_from_heads(loc, scale_raw, sample_shape=torch.Size(), row_offset=0):
    """(z, kappa) straight from the two encoder heads: loc is the un-normalised output of the mean head (B, 3),
    scale_raw the raw output of the concentration head; kappa = softplus(scale_raw) + 1 (model/spherical_vae.py:68-72,
    model/torus_vae.py:101-105, CNN_experiment/circular_VAE.py:47-51) is evaluated inside the sampler kernel and
    returned next to z = VonMisesFisher3D(loc, kappa).rsample(sample_shape).  Same noise as rsample() (u then eps from
    torch's generator, or the in-kernel streams when VMF_RNG=philox, numbered from the global `row_offset`); both outputs
    carry gradients to both heads."""
    sample_shape = torch.Size(sample_shape)
    assert loc.shape[-1] == 3, "This simple version o"
    scale_raw = scale_raw if scale_raw.dtype == loc.dtype else scale_raw.to(loc.dtype)
    _lib.require_hip(loc, scale_raw)
    plan = vmf3_broadcast_plan(loc.shape, scale_raw.shape, sample_shape)
    if rng.rng_mode() == "torch":
        u = torch.rand(plan["w_shape"], device=loc.device, dtype=loc.dtype)          # :56
        eps = torch.randn(plan["e_shape"] + (3,), device=loc.device, dtype=loc.dtype)  # :75
        seed, offset, offset_dev = 0, 0, None
    else:
        u = eps = None
        seed, offset, offset_dev = rng.default_generator.ticket(loc.device)
    return _VMF3RSample.apply(loc, scale_raw, u, eps, plan, seed, offset, offset_dev, True, int(row_offset))


class _VMF3SampleW(torch.autograd.Function):
    """w = 1 + logsumexp([log u, log(1-u) - 2k]) / k elementwise over sample_shape + scale.shape, gradient to scale"""

    @staticmethod
    def forward(ctx, scale, u, w_shape, seed, offset, offset_dev=None, row0=0):
        dev = _lib.require_hip(scale, u)
        sc = scale.detach().contiguous()
        uc = None if u is None else u.to(sc.dtype).contiguous()
        w = torch.empty(w_shape, device=dev, dtype=sc.dtype)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf3_sample_w_fwd(_lib.ptr(sc), _lib.ptr(uc), _lib.ptr(w), w.numel(), sc.numel(),
                                              _lib.dtype_code(sc), row0, seed, offset, _lib.ptr(offset_dev),
                                              _lib.stream_ptr(dev))
        _lib.check(rc, "vmf3_sample_w_fwd")
        ctx.save_for_backward(sc, uc)
        ctx.meta = (row0, seed, offset, offset_dev, tuple(scale.shape))
        return w

    @staticmethod
    def backward(ctx, gw):
        sc, uc = ctx.saved_tensors
        row0, seed, offset, offset_dev, sshape = ctx.meta
        dev = sc.device
        g = gw.to(sc.dtype).contiguous()
        rows = torch.empty_like(g)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf3_sample_w_bwd(_lib.ptr(sc), _lib.ptr(uc), _lib.ptr(g), _lib.ptr(rows), g.numel(),
                                              sc.numel(), _lib.dtype_code(sc), row0, seed, offset, _lib.ptr(offset_dev),
                                              _lib.stream_ptr(dev))
        _lib.check(rc, "vmf3_sample_w_bwd")
        return rows.reshape((-1,) + sshape).sum(0), None, None, None, None, None, None


class VonMisesFisher3D(Distribution):
    """
    Simpler von Mises-Fisher distribution on the unit sphere in R^3 (sampler/von_mises_fisher.py:7-86).

    Parameters
    ----------
    loc : tensor (..., 3)
        Mean direction; normalised as loc / (||loc|| + 1e-5) like the reference (:30).
    scale : tensor (...) -- or (..., 1), which broadcasts rsample() to (B, B, 3) exactly as the reference does.
        Concentration kappa > 0.
    row_offset : int, keyword only (not in the reference, which runs on one device)
        Global index of this tensor's first row when the batch is sharded over ranks: with VMF_RNG=philox the in-kernel
        noise of row i is keyed by row_offset + i, so a rank that holds rows [a, b) and passes row_offset=a draws exactly
        the rows the unsharded call draws ((B,) and (B, 1) scale).  Ignored with torch noise.
    """

    arg_constraints = {
        "loc": constraints.real,
        "scale": constraints.positive,
    }  # type: ignore
    support = constraints.real  # type: ignore
    has_rsample = True

    def __init__(self, loc, scale, validate_args=None, *, row_offset=0):
        assert loc.shape[-1] == 3, "This simple version o"
        self.row_offset = int(row_offset)
        self._loc_raw = loc
        self._loc_n = None
        self.scale = scale
        self.device = loc.device
        self.dtype = loc.dtype
        super().__init__(batch_shape=loc.shape[:-1], validate_args=_capture_safe(validate_args, loc))

    @property
    def loc(self):
        """loc / (||loc|| + 1e-5)  (:30).  The kernels normalise the raw loc themselves, so this tensor is only
        built when somebody reads it (argument validation, mean())."""
        if self._loc_n is None:
            self._loc_n = self._loc_raw / (self._loc_raw.norm(dim=-1, keepdim=True) + 1e-5)
        return self._loc_n

    def mean(self):
        # E[X] = A_m(kappa) * mu with m=3  (:38-43); a method, as in the reference
        m = 3
        kappa = self.scale
        A = ive(m / 2, kappa) / ive(m / 2 - 1, kappa)  # type: ignore
        return self.loc * A.unsqueeze(-1)

    def stddev(self):
        return self.scale  # (:46-48)

    def _sample_w3(self, sample_shape):
        """w of shape sample_shape + scale.shape by the exact inverse CDF of the m = 3 cosine marginal (:50-64);
        one uniform draw (torch.rand, as the reference, or the in-kernel stream when VMF_RNG=philox)."""
        shape = torch.Size(sample_shape) + self.scale.shape
        _lib.require_hip(self.scale)
        if rng.rng_mode() == "torch":
            u = torch.rand(shape, device=self.device, dtype=self.dtype)  # :56
            seed, offset, offset_dev = 0, 0, None
        else:
            u = None
            seed, offset, offset_dev = rng.default_generator.ticket(self.scale.device)
        return _VMF3SampleW.apply(self.scale, u, tuple(shape), seed, offset, offset_dev, self.row_offset)

    def rsample(self, sample_shape=torch.Size()):
        """Reparameterised sample of shape broadcast(sample_shape + scale.shape, sample_shape + batch_shape) + (3,)
        (:66-86).  Noise: VMF_RNG=torch (default) draws u then eps from torch's generator in the reference's
        order; VMF_RNG=philox generates both inside the kernel."""
        sample_shape = torch.Size(sample_shape)
        loc = self._loc_raw
        scale = self.scale if self.scale.dtype == loc.dtype else self.scale.to(loc.dtype)
        _lib.require_hip(loc, scale)
        plan = vmf3_broadcast_plan(loc.shape, scale.shape, sample_shape)
        if rng.rng_mode() == "torch":
            u = torch.rand(plan["w_shape"], device=self.device, dtype=self.dtype)         # :56
            eps = torch.randn(plan["e_shape"] + (3,), device=self.device, dtype=self.dtype)  # :75
            seed, offset, offset_dev = 0, 0, None
        else:
            u = eps = None
            seed, offset, offset_dev = rng.default_generator.ticket(loc.device)
        return _VMF3RSample.apply(loc, scale, u, eps, plan, seed, offset, offset_dev, False, self.row_offset)

    def rsample_with_noise(self, u, eps, sample_shape=torch.Size()):
        """rsample with the caller's noise: u of shape sample_shape + scale.shape (uniform), eps of shape
        sample_shape + loc.shape (normal).  This is what the parity tests drive."""
        sample_shape = torch.Size(sample_shape)
        loc = self._loc_raw
        scale = self.scale if self.scale.dtype == loc.dtype else self.scale.to(loc.dtype)
        plan = vmf3_broadcast_plan(loc.shape, scale.shape, sample_shape)
        if tuple(u.shape) != plan["w_shape"] or tuple(eps.shape) != plan["e_shape"] + (3,):
            raise ValueError(f"noise shapes {tuple(u.shape)}, {tuple(eps.shape)} do not match "
                             f"{plan['w_shape']}, {plan['e_shape'] + (3,)}")
        return _VMF3RSample.apply(loc, scale, u.to(loc.dtype), eps.to(loc.dtype), plan, 0, 0)

    def mean_fused(self):
        """mean() in one kernel (no autograd): loc_n * ive(3/2, k) / ive(1/2, k)."""
        loc = self._loc_raw.detach().contiguous()
        scale = self.scale.detach().to(loc.dtype).expand(loc.shape[:-1]).contiguous()
        dev = _lib.require_hip(loc, scale)
        out = torch.empty_like(loc)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf3_mean(_lib.ptr(loc), _lib.ptr(scale), _lib.ptr(out), scale.numel(),
                                      _lib.dtype_code(loc), _lib.stream_ptr(dev))
        _lib.check(rc, "vmf3_mean")
        return out


class _VMF3LogProbRef(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loc, scale, x):
        dev = _lib.require_hip(loc, scale, x)
        locc, sc, xc = loc.detach().contiguous(), scale.detach().contiguous(), x.detach().contiguous()
        logp = torch.empty_like(sc)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf3_log_prob_ref(_lib.ptr(locc), _lib.ptr(sc), _lib.ptr(xc), _lib.ptr(logp), None, sc.numel(),
                                              _lib.dtype_code(locc), _lib.stream_ptr(dev))
        _lib.check(rc, "vmf3_log_prob_ref")
        ctx.save_for_backward(locc, sc, xc)
        return logp

    @staticmethod
    def backward(ctx, gl):
        locc, sc, xc = ctx.saved_tensors
        dev = locc.device
        g = gl.to(locc.dtype).contiguous()
        dloc, dscale, dx = torch.empty_like(locc), torch.empty_like(sc), torch.empty_like(xc)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf3_log_prob_ref_bwd(_lib.ptr(locc), _lib.ptr(sc), _lib.ptr(xc), _lib.ptr(g), _lib.ptr(dloc),
                                                  _lib.ptr(dscale), _lib.ptr(dx), sc.numel(), _lib.dtype_code(locc),
                                                  _lib.stream_ptr(dev))
        _lib.check(rc, "vmf3_log_prob_ref_bwd")
        return dloc, dscale, dx


class VonMisesFisher3DLogProb(VonMisesFisher3D):
    """The copy in CNN_experiment/von_mises_fisher.py, which adds log_prob / _log_normalization (:88-108) with the
    reference's sign convention: log_prob = kappa <loc, x> - (log kappa - log 4 pi - log(sinh kappa + 1e-10)).
    log_prob is differentiable in loc, scale and x, as it is in the reference."""

    def log_prob(self, x):
        loc = self._loc_raw
        bshape = torch.broadcast_shapes(loc.shape[:-1], self.scale.shape, x.shape[:-1])
        locb = loc.expand(bshape + (3,))
        scaleb = self.scale.to(loc.dtype).expand(bshape)
        xb = x.to(loc.dtype).expand(bshape + (3,))
        if locb.numel() == 0:
            return loc.new_empty(bshape)
        return _VMF3LogProbRef.apply(locb, scaleb, xb)

    def _log_normalization(self):
        """log kappa - log 4 pi - log(sinh kappa + 1e-10) (:97-108), differentiable in kappa as in the reference"""
        loc = self._loc_raw.detach()
        bshape = torch.broadcast_shapes(loc.shape[:-1], self.scale.shape)
        scaleb = self.scale.to(loc.dtype).expand(bshape)
        if scaleb.numel() == 0:
            return loc.new_empty(bshape)
        return _VMF3LogNormRef.apply(scaleb)


class _VMF3LogNormRef(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scale):
        dev = _lib.require_hip(scale)
        sc = scale.detach().contiguous()
        lognorm = torch.empty_like(sc)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf3_log_prob_ref(None, _lib.ptr(sc), None, None, _lib.ptr(lognorm), sc.numel(),
                                              _lib.dtype_code(sc), _lib.stream_ptr(dev))
        _lib.check(rc, "vmf3_log_prob_ref")
        ctx.save_for_backward(sc)
        return lognorm

    @staticmethod
    def backward(ctx, g):
        # log_prob = kappa <l, x> - lognorm: with x = 0 the backward kernel's dscale is -g d lognorm / d kappa
        (sc,) = ctx.saved_tensors
        dev = sc.device
        gl = (-g).to(sc.dtype).contiguous()
        x = torch.zeros(sc.shape + (3,), device=dev, dtype=sc.dtype)
        loc = torch.ones_like(x)
        dscale = torch.empty_like(sc)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf3_log_prob_ref_bwd(_lib.ptr(loc), _lib.ptr(sc), _lib.ptr(x), _lib.ptr(gl), None,
                                                  _lib.ptr(dscale), None, sc.numel(), _lib.dtype_code(sc),
                                                  _lib.stream_ptr(dev))
        _lib.check(rc, "vmf3_log_prob_ref_bwd")
        return dscale


# ---- general m ---------------------------------------------------------------------------------------------
class _VMFRSample(torch.autograd.Function):
    """returns z, then log p(z) if want_logp, then kappa if flags has FLAG_SOFTPLUS_KAPPA (`kappa` then holds the raw head
    output and the returned kappa = softplus(raw) + 1 is differentiable through the same backward launch)"""

    @staticmethod
    def forward(ctx, mu, kappa, n, mu_stride, kappa_stride, max_prop, row0, seed, offset, want_logp, offset_dev=None,
                flags=0):
        dev = _lib.require_hip(mu, kappa)
        m = mu.shape[-1]
        muc, kc = mu.detach().contiguous(), kappa.detach().contiguous()
        z = torch.empty((n, m), device=dev, dtype=mu.dtype)
        w = torch.empty((n,), device=dev, dtype=mu.dtype)
        omw = torch.empty((n,), device=dev, dtype=mu.dtype)
        logp = torch.empty((n,), device=dev, dtype=mu.dtype) if want_logp else None
        soft = bool(flags & _lib.FLAG_SOFTPLUS_KAPPA)
        kout = torch.empty((n,), device=dev, dtype=mu.dtype) if soft else None
        with _lib.on_device(dev):
            rc = _lib.lib().vmf_rsample_fwd(_lib.ptr(muc), mu_stride, _lib.ptr(kc), kappa_stride, _lib.ptr(z),
                                            _lib.ptr(w), _lib.ptr(omw), None, _lib.ptr(logp), _lib.ptr(kout), n, m, None,
                                            None, 0, None, max_prop, row0, seed, offset, _lib.ptr(offset_dev), flags,
                                            _lib.dtype_code(muc), _lib.stream_ptr(dev))
        _lib.check(rc, "vmf_rsample_fwd")
        ctx.save_for_backward(muc, kc, w, omw)
        ctx.meta = (n, m, mu_stride, kappa_stride, row0, seed, offset, mu.shape, kappa.shape)
        ctx.offset_dev, ctx.flags, ctx.want_logp = offset_dev, flags, want_logp
        out = (z,)
        if want_logp:
            ctx.mark_non_differentiable(logp)
            out += (logp,)
        if soft:
            out += (kout,)
        return out if len(out) > 1 else z

    @staticmethod
    def backward(ctx, gz, *rest):
        muc, kc, w, omw = ctx.saved_tensors
        n, m, mu_stride, kappa_stride, row0, seed, offset, mu_shape, k_shape = ctx.meta
        dev = muc.device
        gzc = gz.to(muc.dtype).contiguous()
        gk = rest[-1] if (ctx.flags & _lib.FLAG_SOFTPLUS_KAPPA) and rest else None
        gkc = None if gk is None else gk.to(muc.dtype).contiguous()
        dmu = torch.empty((n, m), device=dev, dtype=muc.dtype)
        dk = torch.empty((n,), device=dev, dtype=muc.dtype)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf_rsample_bwd(_lib.ptr(muc), mu_stride, _lib.ptr(kc), kappa_stride, _lib.ptr(w),
                                            _lib.ptr(omw), _lib.ptr(gzc), None, _lib.ptr(gkc), _lib.ptr(dmu), _lib.ptr(dk),
                                            n, m, row0, seed, offset, _lib.ptr(ctx.offset_dev), ctx.flags,
                                            _lib.dtype_code(muc), _lib.stream_ptr(dev))
        _lib.check(rc, "vmf_rsample_bwd")
        if mu_stride == 0:
            dmu = dmu.sum(0, keepdim=True)
        if kappa_stride == 0:
            dk = dk.sum(0, keepdim=True)
        return (dmu.view(mu_shape), dk.view(k_shape)) + (None,) * 10


def vmf_rsample_from_heads(loc_raw, scale_raw, generator=None, row_offset=0, max_proposals=64):
    """(z, kappa) on S^{m-1} straight from the two encoder heads: loc_raw (B, m) is the un-normalised output of the mean
    head, scale_raw (B,) or (B, 1) the raw output of the concentration head.  The sampler kernels normalise the mean
    rows and evaluate kappa = softplus(scale_raw) + 1 (the head of model/spherical_vae.py:68-72) themselves, forward and
    backward; kappa comes back for the KL term with its gradient routed through the same backward launch.  One sample
    per row (z is (B, m))."""
    assert loc_raw.dim() == 2 and loc_raw.shape[-1] >= 2, "vmf_rsample_from_heads: loc_raw must be (B, m), m >= 2"
    n, m = loc_raw.shape
    raw = scale_raw.to(loc_raw.dtype).reshape(n)
    gen = generator or rng.default_generator
    seed, offset, offset_dev = gen.ticket(loc_raw.device)
    if n == 0:
        return loc_raw.new_empty((0, m)), raw.new_empty((0,)).view(scale_raw.shape)
    z, kappa = _VMFRSample.apply(loc_raw, raw, n, m, 1, int(max_proposals), int(row_offset), seed, offset, False,
                                 offset_dev, _lib.FLAG_NORMALIZE_MU | _lib.FLAG_SOFTPLUS_KAPPA)
    return z, kappa.view(scale_raw.shape)


class _VMFLogProb(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, kappa, x):
        dev = _lib.require_hip(mu, kappa, x)
        n, m = x.shape
        muc, kc, xc = mu.detach().contiguous(), kappa.detach().contiguous(), x.detach().contiguous()
        out = torch.empty((n,), device=dev, dtype=mu.dtype)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf_log_prob_fwd(_lib.ptr(muc), m, _lib.ptr(kc), 1, _lib.ptr(xc), _lib.ptr(out), n, m,
                                             _lib.dtype_code(muc), _lib.stream_ptr(dev))
        _lib.check(rc, "vmf_log_prob_fwd")
        ctx.save_for_backward(muc, kc, xc)
        return out

    @staticmethod
    def backward(ctx, gl):
        muc, kc, xc = ctx.saved_tensors
        n, m = xc.shape
        dev = muc.device
        glc = gl.to(muc.dtype).contiguous()
        dmu, dx, dk = torch.empty_like(muc), torch.empty_like(xc), torch.empty_like(kc)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf_log_prob_bwd(_lib.ptr(muc), m, _lib.ptr(kc), 1, _lib.ptr(xc), _lib.ptr(glc),
                                             _lib.ptr(dmu), _lib.ptr(dx), _lib.ptr(dk), n, m, _lib.dtype_code(muc),
                                             _lib.stream_ptr(dev))
        _lib.check(rc, "vmf_log_prob_bwd")
        return dmu, dk, dx


class _VMFEntropy(torch.autograd.Function):
    """returns (entropy, KL vs uniform); d KL / d kappa = -d H / d kappa = kappa A_m'(kappa)"""

    @staticmethod
    def forward(ctx, kappa, m):
        dev = _lib.require_hip(kappa)
        kc = kappa.detach().contiguous()
        H, kl, dkl = torch.empty_like(kc), torch.empty_like(kc), torch.empty_like(kc)
        with _lib.on_device(dev):
            rc = _lib.lib().vmf_entropy_kl_fwd(_lib.ptr(kc), _lib.ptr(H), _lib.ptr(kl), _lib.ptr(dkl), kc.numel(), m,
                                               _lib.dtype_code(kc), _lib.stream_ptr(dev))
        _lib.check(rc, "vmf_entropy_kl_fwd")
        ctx.save_for_backward(dkl)
        return H, kl

    @staticmethod
    def backward(ctx, gH, gkl):
        (dkl,) = ctx.saved_tensors
        return (gkl - gH) * dkl, None


class VonMisesFisher(Distribution):
    """von Mises-Fisher distribution on S^{m-1} for any m >= 2 (no reference counterpart: the reference asserts
    m == 3, sampler/von_mises_fisher.py:28).

    loc (..., m) is the mean direction (expected unit-norm), scale (...) or (..., 1) the concentration.
    With `normalize_loc=True` loc may be any non-zero vectors (e.g. the raw output of a linear layer): the sampler
    kernels use loc / ||loc|| and return the gradient with respect to the raw rows, so no separate normalisation
    pass (and no pass for its backward) runs over the (n, m) tensor; `.loc` then is the normalised tensor, built
    lazily with torch ops only if log_prob / mean / argument validation ask for it.
    rsample: Wood's rejection sampler for w = <z, e1>, a uniform tangent direction from m-1 Gaussians, and the
    Householder reflection that moves e1 onto loc; all noise is Philox4x32-10 generated inside the kernel and
    keyed by the global row index (`row_offset` + row), so row-sharded ranks reproduce the unsharded draw."""

    arg_constraints = {"loc": constraints.real, "scale": constraints.positive}  # type: ignore
    support = constraints.real  # type: ignore
    has_rsample = True

    def __init__(self, loc, scale, validate_args=None, max_proposals=64, row_offset=0, generator=None,
                 normalize_loc=False):
        assert loc.shape[-1] >= 2, "VonMisesFisher needs an ambient dimension of at least 2."
        self.normalize_loc = bool(normalize_loc)
        self._loc_given = loc
        self._loc_unit = None if self.normalize_loc else loc
        self.scale = scale
        self.device = loc.device
        self.dtype = loc.dtype
        self.m = loc.shape[-1]
        self.max_proposals = int(max_proposals)
        self.row_offset = int(row_offset)
        self.generator = generator or rng.default_generator
        super().__init__(batch_shape=loc.shape[:-1], event_shape=loc.shape[-1:],
                         validate_args=_capture_safe(validate_args, loc))

    @property
    def loc(self):
        if self._loc_unit is None:
            self._loc_unit = torch.nn.functional.normalize(self._loc_given, dim=-1, eps=0.0)
        return self._loc_unit

    def _kappa_rows(self):
        """kappa per batch element.  scale must have the batch shape, the batch shape + (1,) (what a linear head with
        one output gives), or be broadcastable to the batch shape without a trailing 1 that could be taken for either"""
        k = self.scale.to(self.dtype)
        bs = tuple(self.batch_shape)
        if tuple(k.shape) == bs + (1,):
            k = k.squeeze(-1)
        elif k.dim() == len(bs) + 1:
            raise ValueError(f"VonMisesFisher: scale of shape {tuple(k.shape)} does not match batch shape {bs} "
                             f"(expected {bs} or {bs + (1,)})")
        return k.expand(self.batch_shape).reshape(-1)

    def rsample(self, sample_shape=torch.Size(), return_log_prob=False):
        """return_log_prob=True also returns log p(z), fused into the sampler kernel: a VALUE (detached) -- for a term
        that must carry gradients (an entropy estimate in a loss) call log_prob(z), which is differentiable"""
        sample_shape = torch.Size(sample_shape)
        nb, S = _numel(self.batch_shape), _numel(sample_shape)
        n = nb * S
        mu2d = self._loc_given.reshape(nb, self.m)  # raw rows when normalize_loc: the kernel normalises
        kap = self._kappa_rows()
        if nb == 1 and S > 1:
            mu_stride, kappa_stride = 0, 0          # one shared mean: nothing is replicated in memory
        else:
            mu_stride, kappa_stride = self.m, 1
            if S > 1:
                mu2d = mu2d.unsqueeze(0).expand(S, nb, self.m).reshape(n, self.m)
                kap = kap.unsqueeze(0).expand(S, nb).reshape(n)
        seed, offset, offset_dev = self.generator.ticket(self.device)
        out_shape = tuple(sample_shape) + tuple(self.batch_shape)
        if n == 0:
            z = torch.empty(out_shape + (self.m,), device=self.device, dtype=self.dtype)
            return (z, z.new_empty(out_shape)) if return_log_prob else z
        res = _VMFRSample.apply(mu2d, kap, n, mu_stride, kappa_stride, self.max_proposals, self.row_offset, seed,
                                offset, bool(return_log_prob), offset_dev,
                                _lib.FLAG_NORMALIZE_MU if self.normalize_loc else 0)
        if return_log_prob:
            z, logp = res
            return z.view(out_shape + (self.m,)), logp.view(out_shape)
        return res.view(out_shape + (self.m,))

    def sample(self, sample_shape=torch.Size()):
        with torch.no_grad():
            return self.rsample(sample_shape)

    def log_prob(self, x):
        """kappa <loc, x> + log C_m(kappa)"""
        bshape = torch.broadcast_shapes(x.shape[:-1], self.batch_shape)
        n = _numel(bshape)
        xb = x.to(self.dtype).expand(bshape + (self.m,)).reshape(n, self.m)
        mub = self.loc.expand(bshape + (self.m,)).reshape(n, self.m)
        k = self.scale.to(self.dtype)
        if k.dim() == self._loc_given.dim() and k.shape[-1] == 1:
            k = k.squeeze(-1)
        kb = k.expand(bshape).reshape(n)
        if n == 0:
            return xb.new_empty(bshape)
        return _VMFLogProb.apply(mub, kb, xb).view(bshape)

    def entropy(self):
        H, _ = _VMFEntropy.apply(self._kappa_rows(), self.m)
        return H.view(self.batch_shape)

    def kl_uniform(self):
        """KL(self || U(S^{m-1})) per batch element, in one kernel"""
        _, kl = _VMFEntropy.apply(self._kappa_rows(), self.m)
        return kl.view(self.batch_shape)

    @property
    def mean(self):
        from .ive import bessel_ratio
        A = bessel_ratio(self.m / 2, self._kappa_rows()).view(self.batch_shape)  # differentiable in kappa
        return self.loc * A.unsqueeze(-1)

    @property
    def stddev(self):
        return self.scale


@register_kl(VonMisesFisher, HypersphericalUniform)
def _kl_vmf_uniform(vmf, hyu):
    assert hyu.dim + 1 == vmf.m, "dimension mismatch between VonMisesFisher and HypersphericalUniform"
    return vmf.kl_uniform()
