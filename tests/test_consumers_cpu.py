"""CPU checks of the steps on either side of the sampler (ELBO, torus embedding, latent losses): the numpy oracle
against the golden vectors, and the library's per-row device math (run on the host through the check library)
against both.  The kernels themselves are covered by the gpu-marked tests."""
import ctypes as C

import numpy as np
import pytest

from oracle import consumers_oracle as co
from test_hostcheck_math import hc, _p  # noqa: F401  (fixture + pointer helper)

TAGS = [("f64", 1e-12, 1e-13), ("f32", 2e-5, 2e-6)]
DATASETS = {"S1": "S1_dataset", "S2": "S2_dataset", "T2": "T2_dataset"}


@pytest.mark.parametrize("tag,rtol,atol", TAGS)
def test_oracle_elbo_matches_reference(golden, tag, rtol, atol):
    g = golden("g_consumers")
    for model, variant, gamma, beta in (("sph", "spherical", 0.7, 0.03), ("cir", "circular", 1.3, 0.2)):
        k = f"{model}_{tag}::"
        val = co.elbo(g[k + "x"], g[k + "x_mu"], g[k + "kappa"], gamma, beta, variant)
        np.testing.assert_allclose(val, g[k + "loss"], rtol=rtol, atol=atol)


@pytest.mark.parametrize("tag,rtol,atol", TAGS)
def test_oracle_build_torus_matches_formula_goldens(golden, tag, rtol, atol):
    g = golden("g_consumers")
    for name in ("bb", "b"):
        k = f"tor_{name}_{tag}::"
        out = co.build_torus(g[k + "zt"], g[k + "zp"], 1.5, 0.5)
        np.testing.assert_allclose(out, g[k + "out"], rtol=rtol, atol=atol)
        gt, gp = co.build_torus_bwd(g[k + "zt"], g[k + "zp"], g[k + "c"], 1.5, 0.5)
        np.testing.assert_allclose(gt, g[k + "g_zt"], rtol=rtol, atol=atol)
        np.testing.assert_allclose(gp, g[k + "g_zp"], rtol=rtol, atol=atol)


@pytest.mark.parametrize("tag,rtol,atol", TAGS)
@pytest.mark.parametrize("ds", list(DATASETS))
def test_oracle_latent_loss_matches_formula_goldens(golden, tag, rtol, atol, ds):
    g = golden("g_consumers")
    for name in ("bb", "b"):
        k = f"ll_{ds}_{name}_{tag}::"
        val = co.latent_loss(g[k + "labels"], g[k + "z"], DATASETS[ds], 1.5, 0.5)
        np.testing.assert_allclose(val, g[k + "loss"], rtol=rtol, atol=atol)
        if tag == "f64":  # the finite-difference gradient pins the autograd gradient of the formula
            gz = co.latent_loss_bwd(g[k + "labels"], g[k + "z"], DATASETS[ds], 1.5, 0.5)
            np.testing.assert_allclose(gz, g[k + "g_z"], rtol=2e-6, atol=1e-8)


@pytest.mark.parametrize("tag,rtol,atol", TAGS)
@pytest.mark.parametrize("ds", list(DATASETS))
def test_device_latent_loss_math(hc, golden, tag, rtol, atol, ds):  # noqa: F811
    g = golden("g_consumers")
    dt = np.float64 if tag == "f64" else np.float32
    fn = hc.hc_latent_f64 if tag == "f64" else hc.hc_latent_f32
    ct = C.c_double if tag == "f64" else C.c_float
    for name in ("bb", "b"):
        k = f"ll_{ds}_{name}_{tag}::"
        z = np.ascontiguousarray(g[k + "z"], dtype=dt)
        lab = np.ascontiguousarray(g[k + "labels"], dtype=dt)
        n = z.size // 3
        rows, dz = np.empty(n, dt), np.empty((n, 3), dt)
        fn(_p(z), _p(lab), _p(rows), _p(dz), C.c_int64(n), 3, C.c_int64(lab.shape[0]), lab.shape[1],
           list(DATASETS).index(ds), ct(1.5), ct(0.5))
        ref_rows = co.latent_loss_rows(lab, z, DATASETS[ds], 1.5, 0.5).reshape(-1)
        np.testing.assert_allclose(rows, ref_rows, rtol=rtol * 10, atol=atol * 10)
        np.testing.assert_allclose(rows.astype(np.float64).mean(), g[k + "loss"], rtol=rtol, atol=atol)
        # gradient of the mean: analytic per-row derivative / n against torch autograd of the formula
        np.testing.assert_allclose(dz.reshape(z.shape) / n, g[k + "g_z"], rtol=50 * rtol,
                                   atol=(1e-12 if tag == "f64" else 3e-6) / n * 10)


@pytest.mark.parametrize("tag,rtol,atol", TAGS)
def test_device_torus_math(hc, golden, tag, rtol, atol):  # noqa: F811
    g = golden("g_consumers")
    dt = np.float64 if tag == "f64" else np.float32
    fn = hc.hc_torus_f64 if tag == "f64" else hc.hc_torus_f32
    ct = C.c_double if tag == "f64" else C.c_float
    for name in ("bb", "b"):
        k = f"tor_{name}_{tag}::"
        zt, zp, c = (np.ascontiguousarray(g[k + s], dtype=dt) for s in ("zt", "zp", "c"))
        n = zt.size // 3
        out, gt, gp = np.empty((n, 3), dt), np.empty((n, 2), dt), np.empty((n, 2), dt)
        fn(_p(zt), _p(zp), _p(c), _p(out), _p(gt), _p(gp), C.c_int64(n), 3, ct(1.5), ct(0.5))
        np.testing.assert_allclose(out.reshape(g[k + "out"].shape), g[k + "out"], rtol=rtol, atol=atol)
        np.testing.assert_allclose(gt, g[k + "g_zt"].reshape(n, 3)[:, :2], rtol=rtol, atol=atol)
        np.testing.assert_allclose(gp, g[k + "g_zp"].reshape(n, 3)[:, :2], rtol=rtol, atol=atol)
        assert not g[k + "g_zt"].reshape(n, 3)[:, 2].any() and not g[k + "g_zp"].reshape(n, 3)[:, 2].any()


def test_latent_loss_unknown_dataset():
    with pytest.raises(ValueError):
        co.latent_loss_rows(np.zeros((2, 2)), np.ones((2, 3)), "points_dataset")
