"""medicalimageanalysis_amd -- the von Mises-Fisher latent-sampling path of KHOUTAIBI/MedicalImageAnalysis,
rebuilt for MI355X (gfx950): hand-written HIP kernels in libvmf_hip.so behind the reference's Python names.

    from medicalimageanalysis_amd.sampler.von_mises_fisher import VonMisesFisher3D, VonMisesFisher
    from medicalimageanalysis_amd.sampler.ive import ive, Ive, IveFunction
    from medicalimageanalysis_amd.sampler.spherical_uniform import SphericalUniform, HypersphericalUniform
    from medicalimageanalysis_amd.model.spherical_vae import SphericalVAE
    from medicalimageanalysis_amd.model.torus_vae import ToroidalVAE

`install_aliases()` additionally registers the reference's top-level module names (`sampler.*`, `model.*`) so
that code written against the reference imports this implementation unchanged.
"""
import importlib
import sys

from . import _lib, rng  # noqa: F401
from .rng import PhiloxGenerator, manual_seed  # noqa: F401

__version__ = "0.1.0"

_ALIASES = {
    "sampler": "medicalimageanalysis_amd.sampler",
    "sampler.ive": "medicalimageanalysis_amd.sampler.ive",
    "sampler.von_mises_fisher": "medicalimageanalysis_amd.sampler.von_mises_fisher",
    "sampler.spherical_uniform": "medicalimageanalysis_amd.sampler.spherical_uniform",
    "model": "medicalimageanalysis_amd.model",
    "model.spherical_vae": "medicalimageanalysis_amd.model.spherical_vae",
    "model.torus_vae": "medicalimageanalysis_amd.model.torus_vae",
}


def install_aliases():
    """Make `import sampler.von_mises_fisher`, `from model.spherical_vae import *` ... resolve to this package."""
    for alias, target in _ALIASES.items():
        sys.modules[alias] = importlib.import_module(target)
