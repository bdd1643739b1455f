"""ctypes front end of oracle/libvmf_oracle.so (numpy in, numpy out).

TEST INFRASTRUCTURE ONLY: imported by tests/, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``medicalimageanalysis_amd/`` imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvmf_oracle.so")

KL_SPHERICAL_PIECEWISE, KL_TORUS_RAW, KL_CIRCULAR_EPS, KL_S1_I0I1, KL_STABLE_M3, KL_STABLE_M2 = range(6)


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_ive_scalar.restype = C.c_double
        _lib.orc_ive_scalar.argtypes = [C.c_double, C.c_double]
        _lib.orc_log_ive_scalar.restype = C.c_double
        _lib.orc_log_ive_scalar.argtypes = [C.c_double, C.c_double]
        _lib.orc_bessel_ratio_scalar.restype = C.c_double
        _lib.orc_bessel_ratio_scalar.argtypes = [C.c_double, C.c_double]
        _lib.orc_vmf_log_norm_scalar.restype = C.c_double
        _lib.orc_vmf_log_norm_scalar.argtypes = [C.c_int, C.c_double]
        _lib.orc_u01.restype = C.c_double
        _lib.orc_u01.argtypes = [C.c_uint32]
    return _lib


def _suf(dt):
    dt = np.dtype(dt)
    if dt == np.float64:
        return "_f64"
    if dt == np.float32:
        return "_f32"
    raise TypeError(dt)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt=None):
    return np.ascontiguousarray(a, dtype=dt)


def _i64(seq):
    return (C.c_int64 * len(seq))(*[int(s) for s in seq])


def _chk(rc):
    if rc != 0:
        raise RuntimeError(f"oracle call failed rc={rc}")


# ---- broadcast plan of VonMisesFisher3D.rsample, derived with numpy's own broadcasting ----------
def vmf3_plan(loc_shape, scale_shape, sample_shape=()):
    """Shapes and element strides (0 on broadcast dims) of every operand of rsample's output rows."""
    sample_shape = tuple(sample_shape)
    lrows = tuple(loc_shape[:-1])
    w_shape = sample_shape + tuple(scale_shape)
    e_shape = sample_shape + lrows
    o_shape = np.broadcast_shapes(w_shape, e_shape)

    def strides(shape, pad_front_to=None):
        a = np.empty(shape if len(shape) else (), dtype=np.int8)
        b = np.broadcast_to(a, o_shape)
        return [int(s) for s in b.strides]

    w_str = strides(w_shape)
    e_str = strides(e_shape)
    # scale / loc rows do not vary along the sample dims
    k_full = np.broadcast_to(np.empty(tuple(scale_shape), dtype=np.int8), w_shape)
    k_str = [int(s) for s in np.broadcast_to(k_full, o_shape).strides]
    l_full = np.broadcast_to(np.empty(lrows, dtype=np.int8), e_shape)
    l_str = [int(s) for s in np.broadcast_to(l_full, o_shape).strides]
    return dict(o_shape=tuple(o_shape), w_shape=w_shape, e_shape=e_shape, w_str=w_str, k_str=k_str, l_str=l_str,
                e_str=e_str)


def vmf3_rsample_fwd(loc, scale, u, eps, sample_shape=()):
    dt = loc.dtype
    plan = vmf3_plan(loc.shape, scale.shape, sample_shape)
    assert tuple(u.shape) == plan["w_shape"] and tuple(eps.shape) == plan["e_shape"] + (3,)
    loc, scale, u, eps = _c(loc), _c(scale, dt), _c(u, dt), _c(eps, dt)
    z = np.empty(plan["o_shape"] + (3,), dtype=dt)
    w = np.empty(plan["w_shape"], dtype=dt)
    nd = len(plan["o_shape"])
    f = getattr(lib(), "orc_vmf3_rsample_fwd" + _suf(dt))
    _chk(f(_p(loc), _p(scale), _p(u), _p(eps), _p(z), _p(w), nd, _i64(plan["o_shape"]), _i64(plan["w_str"]),
           _i64(plan["k_str"]), _i64(plan["l_str"]), _i64(plan["e_str"])))
    return z, w


def vmf3_rsample_bwd(loc, scale, u, eps, grad_z, sample_shape=()):
    dt = loc.dtype
    plan = vmf3_plan(loc.shape, scale.shape, sample_shape)
    loc, scale, u, eps, grad_z = _c(loc), _c(scale, dt), _c(u, dt), _c(eps, dt), _c(grad_z, dt)
    assert tuple(grad_z.shape) == plan["o_shape"] + (3,)
    dloc = np.empty_like(loc)
    dscale = np.empty_like(scale)
    nd = len(plan["o_shape"])
    f = getattr(lib(), "orc_vmf3_rsample_bwd" + _suf(dt))
    _chk(f(_p(loc), _p(scale), _p(u), _p(eps), _p(grad_z), _p(dloc), _p(dscale), C.c_int64(loc.size // 3),
           C.c_int64(scale.size), nd, _i64(plan["o_shape"]), _i64(plan["w_str"]), _i64(plan["k_str"]),
           _i64(plan["l_str"]), _i64(plan["e_str"])))
    return dloc, dscale


def vmf_kl(kappa, variant):
    """returns (per-element KL, d KL / d kappa, sum in double)"""
    dt = kappa.dtype
    k = _c(kappa)
    out, dout = np.empty_like(k), np.empty_like(k)
    s = C.c_double(0)
    f = getattr(lib(), "orc_vmf_kl" + _suf(dt))
    _chk(f(_p(k), _p(out), _p(dout), C.byref(s), C.c_int64(k.size), int(variant)))
    return out, dout, s.value


def vmf3_log_prob_ref(loc, scale, x):
    dt = loc.dtype
    loc, scale, x = _c(loc), _c(scale, dt), _c(x, dt)
    lp, ln = np.empty_like(scale), np.empty_like(scale)
    f = getattr(lib(), "orc_vmf3_log_prob_ref" + _suf(dt))
    _chk(f(_p(loc), _p(scale), _p(x), _p(lp), _p(ln), C.c_int64(scale.size)))
    return lp, ln


def vmf3_mean(loc, scale):
    dt = loc.dtype
    loc, scale = _c(loc), _c(scale, dt)
    out = np.empty_like(loc)
    _chk(getattr(lib(), "orc_vmf3_mean" + _suf(dt))(_p(loc), _p(scale), _p(out), C.c_int64(scale.size)))
    return out


def sphere_normalise(g):
    g = _c(g)
    out = np.empty_like(g)
    dim = g.shape[-1]
    _chk(getattr(lib(), "orc_sphere_normalise" + _suf(g.dtype))(_p(g), _p(out), C.c_int64(g.size // dim), dim))
    return out


def s1_projnormal_fwd(mean, kappa, noise):
    dt = mean.dtype
    mean, kappa, noise = _c(mean), _c(kappa, dt).reshape(-1), _c(noise, dt)
    z = np.empty_like(mean)
    dim = mean.shape[-1]
    _chk(getattr(lib(), "orc_s1_projnormal_fwd" + _suf(dt))(_p(mean), _p(kappa), _p(noise), _p(z),
                                                            C.c_int64(mean.size // dim), dim))
    return z


def s1_projnormal_bwd(mean, kappa, noise, gz):
    dt = mean.dtype
    mean, kappa, noise, gz = _c(mean), _c(kappa, dt).reshape(-1), _c(noise, dt), _c(gz, dt)
    dm, dk = np.empty_like(mean), np.empty_like(kappa)
    dim = mean.shape[-1]
    _chk(getattr(lib(), "orc_s1_projnormal_bwd" + _suf(dt))(_p(mean), _p(kappa), _p(noise), _p(gz), _p(dm), _p(dk),
                                                            C.c_int64(mean.size // dim), dim))
    return dm, dk


def build_torus(z_theta, z_phi, R_long, r_short):
    dt = z_theta.dtype
    a, b = _c(z_theta), _c(z_phi, dt)
    out = np.empty_like(a)
    f = getattr(lib(), "orc_build_torus" + _suf(dt))
    ft = C.c_double if dt == np.float64 else C.c_float
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, ft, ft]
    _chk(f(_p(a), _p(b), _p(out), a.size // 3, R_long, r_short))
    return out


# ---- ive ------------------------------------------------------------------------------------
def ive(v, z):
    z = _c(z)
    out = np.empty_like(z)
    _chk(getattr(lib(), "orc_ive_fwd" + _suf(z.dtype))(C.c_double(v), _p(z), _p(out), C.c_int64(z.size)))
    return out


def ive_bwd(v, z, gout):
    z = _c(z)
    gout = _c(gout, z.dtype)
    out = np.empty_like(z)
    _chk(getattr(lib(), "orc_ive_bwd" + _suf(z.dtype))(C.c_double(v), _p(z), _p(gout), _p(out), C.c_int64(z.size)))
    return out


def log_ive(v, z):
    z = _c(z, np.float64)
    out = np.empty_like(z)
    _chk(lib().orc_log_ive_f64(C.c_double(v), _p(z), _p(out), C.c_int64(z.size)))
    return out


def bessel_ratio(v, z):
    z = _c(z, np.float64)
    out = np.empty_like(z)
    _chk(lib().orc_bessel_ratio_f64(C.c_double(v), _p(z), _p(out), C.c_int64(z.size)))
    return out


def ive_regime(v, x):
    f = lib().orc_ive_regime
    f.argtypes = [C.c_double, C.c_double]
    return f(v, x)


# ---- general-m sampler -----------------------------------------------------------------------
def philox_raw(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10_raw(c, k, o)
    return [int(x) for x in o]


def vmf_wood_w(kappa, prop_e, prop_u, m):
    dt = kappa.dtype
    kappa, prop_e, prop_u = _c(kappa), _c(prop_e, dt), _c(prop_u, dt)
    n, n_prop = prop_e.shape
    w = np.empty(n, dtype=dt)
    idx = np.empty(n, dtype=np.int32)
    _chk(getattr(lib(), "orc_vmf_wood_w" + _suf(dt))(_p(kappa), _p(prop_e), _p(prop_u), _p(w), _p(idx),
                                                     C.c_int64(n), int(m), int(n_prop)))
    return w, idx


def vmf_rsample_pre(mu, kappa, prop_e, prop_u, gauss):
    dt = mu.dtype
    mu, kappa, prop_e, prop_u, gauss = _c(mu), _c(kappa, dt), _c(prop_e, dt), _c(prop_u, dt), _c(gauss, dt)
    n, m = mu.shape
    n_prop = prop_e.shape[1]
    z = np.empty_like(mu)
    w = np.empty(n, dtype=dt)
    idx = np.empty(n, dtype=np.int32)
    _chk(getattr(lib(), "orc_vmf_rsample_pre" + _suf(dt))(_p(mu), _p(kappa), _p(prop_e), _p(prop_u), _p(gauss), _p(z),
                                                          _p(w), _p(idx), C.c_int64(n), int(m), int(n_prop)))
    return z, w, idx


def vmf_rsample_philox(mu, kappa, seed, offset=0, row0=0, max_prop=64):
    dt = mu.dtype
    mu, kappa = _c(mu), _c(kappa, dt)
    n, m = mu.shape
    z = np.empty_like(mu)
    w = np.empty(n, dtype=dt)
    idx = np.empty(n, dtype=np.int32)
    _chk(getattr(lib(), "orc_vmf_rsample_philox" + _suf(dt))(_p(mu), _p(kappa), _p(z), _p(w), _p(idx), C.c_int64(n),
                                                             int(m), int(max_prop), C.c_uint64(row0),
                                                             C.c_uint64(seed), C.c_uint64(offset)))
    return z, w, idx


def vmf_log_prob(mu, kappa, x):
    dt = mu.dtype
    mu, kappa, x = _c(mu), _c(kappa, dt), _c(x, dt)
    n, m = mu.shape
    out = np.empty(n, dtype=dt)
    _chk(getattr(lib(), "orc_vmf_log_prob" + _suf(dt))(_p(mu), _p(kappa), _p(x), _p(out), C.c_int64(n), int(m)))
    return out


def vmf_entropy_kl(kappa, m):
    k = _c(kappa)
    H, kl = np.empty_like(k), np.empty_like(k)
    _chk(getattr(lib(), "orc_vmf_entropy_kl" + _suf(k.dtype))(_p(k), _p(H), _p(kl), C.c_int64(k.size), int(m)))
    return H, kl
