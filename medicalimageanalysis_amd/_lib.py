"""ctypes binding of libvmf_hip.so (the C ABI declared in include/vmf_hip.h).

There is no CPU fallback: a missing library, a failed call or a tensor that is not on a HIP device
raises.  PyTorch supplies device memory and the current stream; every kernel is in the .so.
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT = os.path.join(_HERE, "libvmf_hip.so")

VMF_F32, VMF_F64 = 0, 1
FLAG_NORMALIZE_MU = 1
FLAG_SOFTPLUS_KAPPA = 2
KIND_GENERAL, KIND_ROWS, KIND_OUTER = 0, 1, 2
KL_SPHERICAL_PIECEWISE, KL_TORUS_RAW, KL_CIRCULAR_EPS, KL_S1_I0I1, KL_STABLE_M3, KL_STABLE_M2 = range(6)
KL_WORKSPACE_DOUBLES = 1024
KL_STATIC_MAX_N = 1 << 30
ELBO_WORKSPACE_DOUBLES = 2048
LATENT_S1, LATENT_S2, LATENT_T2 = range(3)

# (the module object `C` is re-used by callers for byref / c_int64)
_P, _I64, _U64, _I, _D = C.c_void_p, C.c_int64, C.c_uint64, C.c_int, C.c_double
_I64P = C.POINTER(C.c_int64)

# name -> argtypes; every function returns int except the two noted below
_PROTOS = {
    "vmf3_rsample_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I64P, _I64P, _I64P, _I64P, _I64P, _I, _I, _U64, _U64, _U64,
                         _P, _P],
    "vmf3_broadcast_kind": [_I, _I64P, _I64P, _I64P, _I64P, _I64P, _I64P, _I64P],
    "vmf3_rsample_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I, _I64P, _I64P, _I64P, _I64P, _I64P, _I, _I,
                         _U64, _U64, _U64, _P, _P],
    "vmf3_rsample_bwd_outer_workspace": [_I64, _I64, _I, _I64P],
    "vmf3_rsample_bwd_outer": [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _I64, _I, _I, _U64, _U64, _U64, _P, _P],
    "vmf3_sample_w_fwd": [_P, _P, _P, _I64, _I64, _I, _U64, _U64, _U64, _P, _P],
    "vmf3_sample_w_bwd": [_P, _P, _P, _P, _I64, _I64, _I, _U64, _U64, _U64, _P, _P],
    "vmf3_log_prob_ref": [_P, _P, _P, _P, _P, _I64, _I, _P],
    "vmf3_log_prob_ref_bwd": [_P, _P, _P, _P, _P, _P, _P, _I64, _I, _P],
    "vmf3_mean": [_P, _P, _P, _I64, _I, _P],
    "vmf_ive_fwd": [_D, _P, _P, _I64, _I, _P],
    "vmf_ive_bwd": [_D, _P, _P, _P, _I64, _I, _P],
    "vmf_log_ive_fwd": [_D, _P, _P, _I64, _I, _P],
    "vmf_bessel_ratio_fwd": [_D, _P, _P, _I64, _I, _P],
    "vmf_kl_fwd": [_P, _P, _P, _P, _D, _I64, _I, _I, _P],
    "vmf_kl_fwd_static": [_P, _P, _P, _D, _I64, _I, _I, _P],
    "vmf_allreduce_scalars": [_P, _P, _I, _P],
    "vmf_philox_ticket": [_P, _P, _P],
    "vmf_kl_bwd": [_P, _P, _I64, _D, _P, _I64, _I, _I, _P],
    "vmf_rsample_fwd": [_P, _I64, _P, _I64, _P, _P, _P, _P, _P, _P, _I64, _I, _P, _P, _I, _P, _I, _U64, _U64, _U64, _P,
                        _I, _I, _P],
    "vmf_rsample_bwd": [_P, _I64, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _I64, _I, _U64, _U64, _U64, _P, _I, _I, _P],
    "vmf_log_prob_fwd": [_P, _I64, _P, _I64, _P, _P, _I64, _I, _I, _P],
    "vmf_log_prob_bwd": [_P, _I64, _P, _I64, _P, _P, _P, _P, _P, _I64, _I, _I, _P],
    "vmf_entropy_kl_fwd": [_P, _P, _P, _P, _I64, _I, _I, _P],
    "vmf_sphere_uniform_sample": [_P, _P, _I64, _I, _U64, _U64, _U64, _P, _I, _P],
    "vmf_s1_projnormal_fwd": [_P, _P, _P, _P, _I64, _I, _U64, _U64, _U64, _P, _I, _P],
    "vmf_s1_projnormal_bwd": [_P, _P, _P, _P, _P, _P, _I64, _I, _U64, _U64, _U64, _P, _I, _P],
    "vmf_elbo_fwd": [_P, _P, _P, _P, _P, _P, _D, _D, _I64, _I64, _I64, _I, _I, _P],
    "vmf_elbo_bwd": [_P, _P, _P, _P, _P, _P, _D, _D, _I64, _I64, _I64, _I, _I, _P],
    "vmf_build_torus_fwd": [_P, _P, _P, _I64, _I, _D, _D, _I, _P],
    "vmf_build_torus_bwd": [_P, _P, _P, _P, _P, _I64, _I, _D, _D, _I, _P],
    "vmf_latent_loss_fwd": [_P, _P, _P, _P, _P, _I64, _I, _I64, _I, _I, _D, _D, _I, _P],
    "vmf_latent_loss_bwd": [_P, _P, _P, _P, _I64, _I, _I64, _I, _I, _D, _D, _I, _P],
}
EXPORTS = sorted(list(_PROTOS) + ["vmf_version", "vmf_last_error_string"])

_lib = None
_lock = threading.Lock()


def library_path():
    return os.environ.get("VMF_HIP_LIB", _DEFAULT)


def lib():
    """Load the shared library once; raise loudly when it is missing or incomplete."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                path = library_path()
                if not os.path.exists(path):
                    raise RuntimeError(
                        f"libvmf_hip.so not found at {path}: build it with "
                        "`make -C medicalimageanalysis_amd/csrc` (or `python -c 'import __graft_entry__ as g; "
                        "g.build()'`); there is no CPU fallback")
                h = C.CDLL(path)
                for name, argtypes in _PROTOS.items():
                    f = getattr(h, name)  # AttributeError if the symbol is missing
                    f.argtypes = argtypes
                    f.restype = C.c_int
                h.vmf_version.restype = C.c_int
                h.vmf_last_error_string.restype = C.c_char_p
                _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().vmf_last_error_string().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (status {rc}): {msg}")


def dtype_code(t):
    if t.dtype == torch.float32:
        return VMF_F32
    if t.dtype == torch.float64:
        return VMF_F64
    raise TypeError(f"libvmf_hip supports float32 and float64 tensors, got {t.dtype}")


def require_hip(*tensors):
    """Every tensor must live on one HIP device; returns that device."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "medicalimageanalysis_amd runs on MI355X only: got a tensor on "
                f"'{t.device}'. Move the inputs to a HIP device (tensor.to('cuda')); there is no CPU path.")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} and {t.device}")
    return dev


class _Noop:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NOOP = _Noop()


def on_device(dev):
    """Context that makes `dev` the current device for a launch (the library launches on the current device).  When
    it already is -- the usual case -- nothing is switched: torch.cuda.device() costs several microseconds per call,
    a visible share of a launch at the reference's batch sizes."""
    if dev.index is None or torch.cuda.current_device() == dev.index:
        return _NOOP
    return torch.cuda.device(dev)


_workspaces = {}
_retired_workspaces = []


def workspace(dev, n_doubles):
    """Scratch of AT LEAST `n_doubles` float64 for a reduction's block partials (callers pass its real size on): ONE
    buffer per (device, stream), grown in powers of two to the largest request seen, kept for the life of the process.
    Launches on one stream are ordered, so back-to-back calls may share it, and a train step captured into a HIP graph
    contains no allocation for it (the warm-up steps before the capture created it).  A buffer that is outgrown is kept
    alive as well -- a graph captured earlier may still point at it -- which at most doubles the footprint of the
    largest request.  A first call made while the stream is capturing gets a buffer of its own from the capture's pool."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < n_doubles:
        size = 1 << max(10, int(n_doubles - 1).bit_length())
        new = torch.empty(size, device=dev, dtype=torch.float64)
        if torch.cuda.is_current_stream_capturing():
            return new
        if ws is not None:
            _retired_workspaces.append(ws)
        _workspaces[key] = ws = new
    return ws


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def i64_array(seq):
    return (C.c_int64 * len(seq))(*[int(s) for s in seq])
