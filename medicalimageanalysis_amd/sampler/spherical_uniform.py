"""Drop-in for the reference's sampler/spherical_uniform.py, plus `HypersphericalUniform` (log_prob / entropy),
which the reference does not have."""
import math

import torch
from torch.distributions import Distribution, constraints

from .. import _lib, rng


def _sphere_sample(shape_rows, dim, device, dtype, gauss=None, row0=0, generator=None):
    """rows of unit vectors: g / ||g||  (sampler/spherical_uniform.py:31-33)"""
    n = 1
    for s in shape_rows:
        n *= int(s)
    out = torch.empty(tuple(shape_rows) + (dim,), device=device, dtype=dtype)
    dev = _lib.require_hip(out, gauss)
    seed, offset, offset_dev = (0, 0, None)
    if gauss is None:
        seed, offset, offset_dev = (generator or rng.default_generator).ticket(dev)
    with _lib.on_device(dev):
        rc = _lib.lib().vmf_sphere_uniform_sample(_lib.ptr(gauss), _lib.ptr(out), n, dim, row0, seed, offset,
                                                  _lib.ptr(offset_dev),
                                                  _lib.dtype_code(out), _lib.stream_ptr(dev))
    _lib.check(rc, "vmf_sphere_uniform_sample")
    return out


class SphericalUniform(Distribution):
    """Uniform distribution on the unit sphere S^{dim-1} in R^dim  (sampler/spherical_uniform.py:4-33).
    `dim` is the ambient dimension, as in the reference.  Only `sample` exists there; log_prob / entropy raise
    NotImplementedError exactly as the reference's class does."""

    arg_constraints = {}  # type: ignore
    support = constraints.real  # type: ignore
    has_rsample = False

    def __init__(self, dim, validate_args=None, device=None, dtype=torch.float32):
        assert dim >= 2, "Dimension must be at least 2."
        self.dim = dim
        self.device = device
        self.dtype = dtype
        super().__init__(batch_shape=torch.Size(), event_shape=torch.Size([dim]), validate_args=validate_args)

    def sample(self, sample_shape=torch.Size()):
        shape = tuple(sample_shape) + (self.dim,)
        device = torch.device(self.device if self.device is not None else "cuda")
        if device.type != "cuda":
            raise RuntimeError("SphericalUniform.sample runs on MI355X only: construct it with device='cuda'")
        if rng.rng_mode() == "torch":
            z = torch.randn(shape, device=device, dtype=self.dtype)  # the reference's draw (:32)
            return _sphere_sample(shape[:-1], self.dim, device, self.dtype, gauss=z.contiguous())
        return _sphere_sample(shape[:-1], self.dim, device, self.dtype)


class HypersphericalUniform(Distribution):
    """Uniform distribution on S^{dim} embedded in R^{dim+1} (the convention of hyperspherical VAEs: `dim` is the
    manifold dimension).  No reference counterpart."""

    arg_constraints = {}  # type: ignore
    support = constraints.real  # type: ignore
    has_rsample = False

    def __init__(self, dim, validate_args=None, device="cuda", dtype=torch.float32):
        assert dim >= 1, "Manifold dimension must be at least 1."
        self.dim = dim
        self.device = torch.device(device)
        self.dtype = dtype
        super().__init__(batch_shape=torch.Size(), event_shape=torch.Size([dim + 1]), validate_args=validate_args)

    def sample(self, sample_shape=torch.Size(), row_offset=0, generator=None):
        return _sphere_sample(tuple(sample_shape), self.dim + 1, self.device, self.dtype, row0=row_offset,
                              generator=generator)

    def _log_surface_area(self):
        m = self.dim + 1
        return math.log(2.0) + 0.5 * m * math.log(math.pi) - math.lgamma(0.5 * m)

    def entropy(self):
        return torch.tensor(self._log_surface_area(), device=self.device, dtype=self.dtype)

    def log_prob(self, x):
        return -torch.ones(x.shape[:-1], device=x.device, dtype=x.dtype) * self._log_surface_area()
