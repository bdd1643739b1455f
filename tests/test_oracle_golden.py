"""The CPU oracle against the golden vectors produced by the reference (tests/golden/make_golden.py).

This is what pins the oracle: every array compared here came out of the reference's own code
(sampler/von_mises_fisher.py, sampler/ive.py + scipy 1.15.3, model/spherical_vae.py,
CNN_experiment/*.py, regular_cnn/s1_vae.py) run in the build container.
"""
import numpy as np
import pytest

F64_ATOL = 1e-12   # elementwise algebra in fp64: far inside the 1e-6 contract
F32_RTOL = 1e-4    # fp32 contract: 1e-4 relative (to the unit row norm for z)


def sinh_window(k):
    """kappa values for which exp(k) overflows but sinh(k) = e^k/2 does not: there the reference's
    result depends on which sinh torch dispatches to (vectorised sleef overflows early, libm does not),
    i.e. on the element's position in the tensor; not a contract, so those points are not compared."""
    k = np.asarray(k, dtype=np.float64)
    return ((k > 709.78) & (k < 710.48)) | ((k > 88.72) & (k < 89.42))


def _close32(a, b, scale=1.0):
    np.testing.assert_allclose(a, b, rtol=F32_RTOL, atol=F32_RTOL * scale)


@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("seed", range(4))
def test_rsample_elementwise(golden, oracle, tag, seed):
    g = golden("g_rsample")
    p = f"ew_{tag}_s{seed}_"
    z, w = oracle.vmf3_rsample_fwd(g[p + "loc"], g[p + "scale"], g[p + "u"], g[p + "eps"])
    if tag == "f64":
        np.testing.assert_allclose(z, g[p + "z"], rtol=0, atol=F64_ATOL)
        np.testing.assert_allclose(w, g[p + "w"], rtol=0, atol=F64_ATOL)
    else:
        _close32(z, g[p + "z"])
        _close32(w, g[p + "w"])


@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("seed", range(4))
def test_rsample_outer_quirk(golden, oracle, tag, seed):
    """scale (B,1) broadcasts to z (B,B,3): z[i,j] = w[i] loc[j] + f[i] v[j]"""
    g = golden("g_rsample")
    p = f"outer_{tag}_s{seed}_"
    z, _ = oracle.vmf3_rsample_fwd(g[p + "loc"], g[p + "scale"], g[p + "u"], g[p + "eps"])
    assert z.shape == g[p + "z"].shape == (33, 33, 3)
    if tag == "f64":
        np.testing.assert_allclose(z, g[p + "z"], rtol=0, atol=F64_ATOL)
    else:
        _close32(z, g[p + "z"])


@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("case,ss", [("ss5", (5,)), ("ss4o", (4,)), ("nd", ())])
def test_rsample_shapes(golden, oracle, tag, case, ss):
    g = golden("g_rsample")
    p = f"{case}_{tag}_"
    z, _ = oracle.vmf3_rsample_fwd(g[p + "loc"], g[p + "scale"], g[p + "u"], g[p + "eps"], sample_shape=ss)
    assert z.shape == g[p + "z"].shape
    if tag == "f64":
        np.testing.assert_allclose(z, g[p + "z"], rtol=0, atol=F64_ATOL)
    else:
        _close32(z, g[p + "z"])


@pytest.mark.parametrize("case", ["ew", "outer"])
def test_rsample_backward(golden, oracle, case):
    g = golden("g_grads")
    p = case + "_"
    dloc, dscale = oracle.vmf3_rsample_bwd(g[p + "loc"], g[p + "scale"], g[p + "u"], g[p + "eps"], g[p + "c"])
    np.testing.assert_allclose(dloc, g[p + "dloc"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(dscale, g[p + "dscale"], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("name,variant", [("spherical", 0), ("torus", 1), ("circular", 2), ("s1", 3)])
def test_kl_variants(golden, oracle, tag, name, variant):
    g = golden("g_kl")
    k = g[f"kappa_{tag}"]
    ref = g[f"{name}_{tag}"]
    out, _, _ = oracle.vmf_kl(k, variant)
    # NaN / inf positions are part of the contract
    keep = ~sinh_window(k) if tag == "f64" else ~((k > 88.72) & (k < 89.42))
    np.testing.assert_array_equal(np.isnan(out)[keep], np.isnan(ref)[keep])
    np.testing.assert_array_equal(np.isinf(out)[keep], np.isinf(ref)[keep])
    fin = np.isfinite(ref) & keep
    if tag == "f64":
        np.testing.assert_allclose(out[fin], ref[fin], rtol=1e-9, atol=1e-9)
    else:
        np.testing.assert_allclose(out[fin], ref[fin], rtol=F32_RTOL, atol=2e-4)


def test_kl_spherical_grad(golden, oracle):
    g = golden("g_kl")
    k = g["kappa_f64"]
    _, d, _ = oracle.vmf_kl(k, 0)
    ref = g["spherical_grad_f64"]
    np.testing.assert_allclose(d, ref, rtol=1e-9, atol=1e-10)


def test_kl_golden_values_from_survey(oracle):
    """fp64 values measured on the reference, model/spherical_vae.py:141-184"""
    k = np.array([1e-4, .5, 1, 5, 19.9, 20.1, 50, 100, 1e3, 1e4])
    ref = np.array([0, 0.0406518123, 0.1515959042, 1.3030845131, 2.6838669123, 2.6938669956, 3.6051701860,
                    4.2983173665, 6.6009024595, 8.9034875525])
    out, _, s = oracle.vmf_kl(k, 0)
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-9)
    assert abs(s - ref.sum()) < 1e-8


def test_kl_stable_matches_reference_where_finite(golden, oracle):
    g = golden("g_kl")
    k = g["kappa_f64"]
    out, _, _ = oracle.vmf_kl(k, 4)
    ref = g["torus_f64"]
    ok = np.isfinite(ref) & (k > 1e-2)
    np.testing.assert_allclose(out[ok], ref[ok], rtol=1e-9, atol=1e-9)
    assert np.all(np.isfinite(out))
    out2, _, _ = oracle.vmf_kl(k, 5)
    ref2 = g["s1_f64"]
    ok = np.isfinite(ref2)
    np.testing.assert_allclose(out2[ok], ref2[ok], rtol=1e-9, atol=1e-9)
    assert np.all(np.isfinite(out2))


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_s1_loss(golden, oracle, tag):
    g = golden("g_kl")
    kk = g[f"s1_loss_kappa_{tag}"]
    _, _, s = oracle.vmf_kl(kk.reshape(-1), 3)
    mse = 4 * 2 * 0.25
    np.testing.assert_allclose(mse + s, g[f"s1_loss_{tag}"], rtol=1e-6)


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_ive_forward_backward(golden, oracle, tag):
    g = golden("g_ive")
    z = g[f"z_{tag}"]
    for v in g["orders"]:
        v = float(v)
        key = f"{tag}_v{int(v) if v == int(v) else v}"
        ref_f, ref_b = g["fwd_" + key], g["bwd_" + key]
        out = oracle.ive(v, z)
        np.testing.assert_array_equal(np.isnan(out), np.isnan(ref_f), err_msg=f"v={v}")
        fin = np.isfinite(ref_f)
        if tag == "f64":
            np.testing.assert_allclose(out[fin], ref_f[fin], rtol=1e-12, atol=1e-300, err_msg=f"v={v}")
        else:
            np.testing.assert_allclose(out[fin], ref_f[fin], rtol=2e-7, atol=1e-45, err_msg=f"v={v}")
        gz = oracle.ive_bwd(v, z, np.ones_like(z))
        fin = np.isfinite(ref_b)
        np.testing.assert_array_equal(np.isnan(gz), np.isnan(ref_b), err_msg=f"bwd v={v}")
        if tag == "f64":
            np.testing.assert_allclose(gz[fin], ref_b[fin], rtol=1e-9, atol=1e-15, err_msg=f"bwd v={v}")
        else:
            np.testing.assert_allclose(gz[fin], ref_b[fin], rtol=F32_RTOL, atol=1e-6, err_msg=f"bwd v={v}")


def test_ive_spot_values(oracle):
    """values measured on scipy 1.15.3 through sampler/ive.py"""
    lib = oracle.lib()
    assert abs(lib.orc_ive_scalar(255, 1e5) / 9.114020178330799e-4 - 1) < 1e-12
    assert abs(lib.orc_ive_scalar(256, 1e5) / lib.orc_ive_scalar(255, 1e5) / 0.9974482512647274 - 1) < 1e-12
    assert abs(lib.orc_ive_scalar(31, 1e4) / 3.8023020888063815e-3 - 1) < 1e-12
    assert abs(lib.orc_ive_scalar(0.5, 1.0) / 0.3449513138882451 - 1) < 1e-13
    assert lib.orc_ive_scalar(255, 10.0) < 1e-300
    assert np.isnan(lib.orc_ive_scalar(0.5, -1.0))
    assert abs(lib.orc_ive_scalar(1, -1.0) + 0.20791041534970842) < 1e-15
    assert lib.orc_ive_scalar(2.5, 0.0) == 0.0 and lib.orc_ive_scalar(0, 0.0) == 1.0
    assert np.isnan(oracle.ive(0.5, np.array([1e300]))[0])


def test_ive_every_regime_is_exercised(golden, oracle):
    g = golden("g_ive")
    seen = set()
    for v in g["orders"]:
        for x in g["z_f64"]:
            if 0 < x < 1e9:
                seen.add(oracle.ive_regime(float(v), float(x)))
    assert seen == {0, 1, 2, 3}


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_log_prob_reference_sign(golden, oracle, tag):
    g = golden("g_logprob")
    lp, ln = oracle.vmf3_log_prob_ref(g[f"loc_{tag}"], g[f"scale_{tag}"], g[f"x_{tag}"])
    ref, refn = g[f"logprob_{tag}"], g[f"lognorm_{tag}"]
    np.testing.assert_array_equal(np.isinf(lp), np.isinf(ref))
    fin = np.isfinite(ref)
    tol = dict(rtol=1e-12, atol=1e-12) if tag == "f64" else dict(rtol=F32_RTOL, atol=1e-4)
    np.testing.assert_allclose(lp[fin], ref[fin], **tol)
    np.testing.assert_allclose(ln[fin], refn[fin], **tol)


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_mean(golden, oracle, tag):
    g = golden("g_mean")
    out = oracle.vmf3_mean(g[f"loc_{tag}"], g[f"scale_{tag}"])
    tol = dict(rtol=1e-12, atol=1e-14) if tag == "f64" else dict(rtol=F32_RTOL, atol=1e-6)
    np.testing.assert_allclose(out, g[f"mean_{tag}"], **tol)
    # closed form A_3(k) = coth k - 1/k
    k = g["scale_f64"]
    A = np.linalg.norm(oracle.vmf3_mean(g["loc_f64"], k), axis=-1) / np.linalg.norm(
        g["loc_f64"] / (np.linalg.norm(g["loc_f64"], axis=-1, keepdims=True) + 1e-5), axis=-1)
    np.testing.assert_allclose(A, 1 / np.tanh(k) - 1 / k, rtol=1e-9)


def test_spherical_uniform(golden, oracle):
    g = golden("g_uniform")
    for d in (2, 3, 10):
        out = oracle.sphere_normalise(g[f"randn_d{d}"])
        np.testing.assert_allclose(out, g[f"sample_d{d}"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_s1_projected_normal(golden, oracle, tag):
    g = golden("g_s1")
    z = oracle.s1_projnormal_fwd(g[f"pn_mean_{tag}"], g[f"pn_kappa_{tag}"], g[f"pn_noise_{tag}"])
    tol = dict(rtol=0, atol=1e-13) if tag == "f64" else dict(rtol=F32_RTOL, atol=1e-6)
    np.testing.assert_allclose(z, g[f"pn_z_{tag}"], **tol)
    z = oracle.s1_projnormal_fwd(g["mean_vec"], g["kappa"], g["noise"])
    np.testing.assert_allclose(z, g["z"], rtol=0, atol=1e-13)


# ---- the torch-op restatement that bench.py times as the CPU baseline of the m = 3 configurations ---------------------
@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("case,ss", [("ew_{}_s0_", ()), ("outer_{}_s1_", ()), ("ss5_{}_", (5,)), ("nd_{}_", ())])
def test_reference_torch_rsample(golden, tag, case, ss):
    import torch
    from oracle import reference_torch as rt

    g = golden("g_rsample")
    p = case.format(tag)
    t = lambda k: torch.from_numpy(g[p + k])  # noqa: E731
    z, _ = rt.rsample(t("loc"), t("scale"), ss, noise=(t("u"), t("eps")))
    assert tuple(z.shape) == g[p + "z"].shape
    if tag == "f64":
        np.testing.assert_allclose(z.numpy(), g[p + "z"], rtol=0, atol=F64_ATOL)
    else:
        _close32(z.numpy(), g[p + "z"])


def test_reference_torch_kl_and_vae_step(golden):
    import torch
    from oracle import reference_torch as rt

    g = golden("g_kl")
    for tag in ("f64", "f32"):
        k = g[f"kappa_{tag}"]
        got = rt.kl_spherical(torch.from_numpy(k)).numpy()
        ok = ~sinh_window(k)
        # fp32: log k - log sinh k cancels below kappa ~ 0.1, libm-to-libm differences are absolute there
        np.testing.assert_allclose(got[ok], g[f"spherical_{tag}"][ok], rtol=1e-12 if tag == "f64" else 1e-4,
                                   atol=1e-12 if tag == "f64" else 2e-4, equal_nan=True)
    g = golden("g_svae")
    cfg = {k[5:]: g[k].item() for k in g.files if k.startswith("cfg::")}
    cfg["device"] = "cpu"
    model = rt.SphericalVAERef(cfg).double()
    model.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w::")}, strict=False)
    x = torch.from_numpy(g["x"])
    z, x_mu, params = model(x, noise=(torch.from_numpy(g["u"]), torch.from_numpy(g["eps"])))
    np.testing.assert_allclose(z.detach().numpy(), g["z"], rtol=0, atol=1e-12)
    loss = model.elbo(x, x_mu, params)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-12)
    loss.backward()
    np.testing.assert_allclose(model.function_z_logvar.weight.grad.numpy(), g["g::function_z_logvar.weight"], rtol=1e-8,
                               atol=1e-12)
