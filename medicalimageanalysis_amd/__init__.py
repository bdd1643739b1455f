"""medicalimageanalysis_amd -- the von Mises-Fisher latent-sampling path of KHOUTAIBI/MedicalImageAnalysis,
rebuilt for MI355X (gfx950): hand-written HIP kernels in libvmf_hip.so behind the reference's Python names.

    from medicalimageanalysis_amd.sampler.von_mises_fisher import VonMisesFisher3D, VonMisesFisher
    from medicalimageanalysis_amd.sampler.ive import ive, Ive, IveFunction
    from medicalimageanalysis_amd.sampler.spherical_uniform import SphericalUniform, HypersphericalUniform
    from medicalimageanalysis_amd.model.spherical_vae import SphericalVAE
    from medicalimageanalysis_amd.model.torus_vae import ToroidalVAE

`install_aliases()` additionally registers the reference's module names for this path (`sampler.*`,
`model.spherical_vae`, `model.torus_vae`; with `flat=True` also the top-level names its CNN / S^1 experiments import)
so that code written against the reference imports this implementation unchanged.
"""
import importlib
import sys
import types

from . import _lib, rng  # noqa: F401
from .rng import PhiloxGenerator, manual_seed  # noqa: F401

__version__ = "0.2.0"

# leaf modules of the reference that this package replaces (sampler/*.py, model/spherical_vae.py, model/torus_vae.py)
_ALIASES = {
    "sampler.ive": "medicalimageanalysis_amd.sampler.ive",
    "sampler.von_mises_fisher": "medicalimageanalysis_amd.sampler.von_mises_fisher",
    "sampler.spherical_uniform": "medicalimageanalysis_amd.sampler.spherical_uniform",
    "model.spherical_vae": "medicalimageanalysis_amd.model.spherical_vae",
    "model.torus_vae": "medicalimageanalysis_amd.model.torus_vae",
}

# The experiments under CNN_experiment/, regular_cnn/ and invariant_cnn/ run with their own directory on sys.path and
# import each other by bare file name (CNN_experiment/circular_VAE.py:5, CNN_experiment/von_mises_fisher.py:3,
# CNN_experiment/train_experiment.py:9, regular_cnn/main.py:11, invariant_cnn/main_invariant.py:10).
_FLAT_ALIASES = {
    "ive": "medicalimageanalysis_amd.sampler.ive",
    "circular_VAE": "medicalimageanalysis_amd.model.circular_vae",
    "s1_vae": "medicalimageanalysis_amd.model.s1_vae",
    "geometry": "medicalimageanalysis_amd.geometry",   # compute_curvature_profile (regular_cnn/main.py:12)
}


def _parent_package(name):
    """The caller's own package of that name when one is importable (the reference's `model/` also holds train.py and
    infer.py, which are not replaced and must keep resolving), else an empty package that only carries the aliases."""
    pkg = sys.modules.get(name)
    if pkg is not None and not getattr(pkg, "_vmf_alias_stub", False):
        return pkg  # the caller's package, already imported: left exactly as it is (never popped, never re-executed)
    if pkg is not None:
        del sys.modules[name]  # an empty stub from an earlier install_aliases(): look again, the path may have changed
    try:
        pkg = importlib.import_module(name)
    except ModuleNotFoundError as e:
        if e.name != name:
            # the caller HAS such a package and its own __init__ failed on a missing dependency: that is the caller's
            # error to see, not something to paper over with an empty package
            raise
        pkg = types.ModuleType(name, "alias package created by medicalimageanalysis_amd.install_aliases()")
        pkg.__path__ = []
        pkg._vmf_alias_stub = True
        sys.modules[name] = pkg
    return pkg


def _flat_von_mises_fisher():
    """CNN_experiment/von_mises_fisher.py as a module: its VonMisesFisher3D is the variant WITH log_prob
    (CNN_experiment/von_mises_fisher.py:88-108), and `from von_mises_fisher import *` hands on the names that file
    imports at its top (:1-4)."""
    src = importlib.import_module("medicalimageanalysis_amd.sampler.von_mises_fisher")
    mod = types.ModuleType("von_mises_fisher", _flat_von_mises_fisher.__doc__)
    for name in ("math", "torch", "ive", "Distribution", "constraints"):
        setattr(mod, name, getattr(src, name))
    mod.VonMisesFisher3D = src.VonMisesFisher3DLogProb
    mod.__all__ = ["math", "torch", "ive", "Distribution", "constraints", "VonMisesFisher3D"]
    return mod


def install_aliases(flat=False):
    """Make `import sampler.von_mises_fisher`, `from model.spherical_vae import *` ... resolve to this package.

    Only the leaf modules this package replaces are registered; `sampler` and `model` themselves stay the caller's
    packages when they exist, so `from model.train import *` (main.py:4) still finds the reference's own train.py.
    flat=True also registers the bare names the CNN / S^1 experiments use (`von_mises_fisher`, `ive`, `circular_VAE`,
    `s1_vae`, `geometry`); they are generic names, hence opt-in."""
    for alias, target in _ALIASES.items():
        parent, leaf = alias.split(".")
        mod = importlib.import_module(target)
        pkg = _parent_package(parent)
        sys.modules[alias] = mod
        setattr(pkg, leaf, mod)
    if flat:
        for alias, target in _FLAT_ALIASES.items():
            sys.modules[alias] = importlib.import_module(target)
        sys.modules["von_mises_fisher"] = _flat_von_mises_fisher()
