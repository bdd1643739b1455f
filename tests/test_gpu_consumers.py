"""HIP kernels for the steps on either side of the sampler (fused ELBO, torus embedding, latent losses) against the
golden vectors, the numpy oracle and a torch-autograd restatement run on the same device.  Needs an MI355X."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a HIP device", allow_module_level=True)

from medicalimageanalysis_amd import _lib, losses  # noqa: E402
from medicalimageanalysis_amd.model.torus_vae import ToroidalVAE  # noqa: E402
from oracle import consumers_oracle as co  # noqa: E402

DEV = torch.device("cuda:0")
TD = {"f64": torch.float64, "f32": torch.float32}
TOL = {"f64": dict(rtol=1e-10, atol=1e-12), "f32": dict(rtol=1e-4, atol=1e-6)}
DATASETS = {"S1": "S1_dataset", "S2": "S2_dataset", "T2": "T2_dataset"}


def T(a, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


# ---- ELBO -------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("model,variant,gamma,beta", [("sph", "spherical", 0.7, 0.03), ("cir", "circular", 1.3, 0.2)])
def test_elbo_matches_reference_values_and_gradients(golden, tag, model, variant, gamma, beta):
    g = golden("g_consumers")
    k = f"{model}_{tag}::"
    x = T(g[k + "x"])
    x_mu = T(g[k + "x_mu"]).requires_grad_()
    kap = T(g[k + "kappa"]).requires_grad_()
    loss = losses.elbo_loss(x, x_mu, kap, gamma, beta, variant)
    assert loss.shape == () and loss.dtype == TD[tag]
    np.testing.assert_allclose(N(loss), g[k + "loss"], **TOL[tag])
    loss.backward()
    np.testing.assert_allclose(N(x_mu.grad), g[k + "g_x_mu"], **TOL[tag])
    gk = g[k + "g_kappa"]
    np.testing.assert_allclose(N(kap.grad), gk, rtol=TOL[tag]["rtol"], atol=TOL[tag]["atol"] + 1e-6 * np.abs(gk).max())


def test_elbo_through_the_c_abi_with_parts():
    """direct call: loss, its two parts, workspace contract"""
    torch.manual_seed(0)
    B, D = 64, 784
    x = torch.rand(B, D, device=DEV)
    xh = torch.rand(B, B, D, device=DEV)
    kap = torch.rand(B, 1, device=DEV) * 50 + 1
    loss = torch.empty((), device=DEV)
    parts = torch.empty(2, device=DEV, dtype=torch.float64)
    ws = torch.empty(_lib.ELBO_WORKSPACE_DOUBLES, device=DEV, dtype=torch.float64)
    rc = _lib.lib().vmf_elbo_fwd(_lib.ptr(x), _lib.ptr(xh), _lib.ptr(kap), _lib.ptr(loss), _lib.ptr(parts),
                                 _lib.ptr(ws), 1.0 / xh.numel(), 0.01 / B, x.numel(), xh.numel(), B,
                                 _lib.KL_SPHERICAL_PIECEWISE, _lib.VMF_F32, _lib.stream_ptr(DEV))
    assert rc == 0
    ref = co.elbo(N(x), N(xh), N(kap), 1.0, 0.01, "spherical")
    np.testing.assert_allclose(N(loss), ref, rtol=2e-6)
    rec64 = float(((x.double() - xh.double()) ** 2).mean())
    np.testing.assert_allclose(N(parts)[0], rec64, rtol=1e-6)      # fp32 squares, double accumulation
    assert abs(N(parts).sum() - float(loss)) < 1e-6
    # n_hat not a multiple of n_x is an argument error, not a launch
    rc = _lib.lib().vmf_elbo_fwd(_lib.ptr(x), _lib.ptr(xh), _lib.ptr(kap), _lib.ptr(loss), None, _lib.ptr(ws), 1.0,
                                 1.0, x.numel(), xh.numel() - 1, B, 0, _lib.VMF_F32, _lib.stream_ptr(DEV))
    assert rc == -1 and b"multiple" in _lib.lib().vmf_last_error_string()


@pytest.mark.parametrize("n_x,reps", [(784 * 128, 420), (100, 400003), (4096, 9001), (4100, 9001)])
def test_elbo_backward_of_large_tensors_one_tile_per_wave(n_x, reps):
    """from 8192 tiles of 16 KiB on, d x_hat is written by the kernel that gives every wave one contiguous tile: same
    values as c (x_hat - x[e mod n_x]) element for element, for x longer and shorter than a tile, a ragged last tile,
    and d kappa from the same launch"""
    gen = torch.Generator(device=DEV).manual_seed(n_x)
    n_hat = n_x * reps
    assert n_hat >= 8192 * 4096
    x = torch.randn(n_x, device=DEV, generator=gen)
    xh = torch.randn(n_hat, device=DEV, generator=gen)
    kap = torch.rand(777, device=DEV, generator=gen) * 40 + 1
    go = torch.tensor([1.7], device=DEV)
    d = torch.full((n_hat + 8,), float("nan"), device=DEV)
    dk = torch.full((777,), float("nan"), device=DEV)
    a_rec, a_kl = 0.3 / n_hat, 0.01 / 777
    rc = _lib.lib().vmf_elbo_bwd(_lib.ptr(x), _lib.ptr(xh), _lib.ptr(kap), _lib.ptr(go), _lib.ptr(d), _lib.ptr(dk), a_rec,
                                 a_kl, n_x, n_hat, 777, _lib.KL_SPHERICAL_PIECEWISE, _lib.VMF_F32, _lib.stream_ptr(DEV))
    assert rc == 0, _lib.lib().vmf_last_error_string()
    c = np.float32(2.0) * np.float32(a_rec) * np.float32(1.7)
    ref = torch.tensor(c, device=DEV) * (xh.view(reps, n_x) - x)
    assert torch.equal(d[:n_hat].view(reps, n_x), ref)
    assert bool(torch.isnan(d[n_hat:]).all())            # nothing past the end
    # d kappa: the grid-stride kernel's values (a call without d x_hat takes it)
    dk2 = torch.empty_like(dk)
    rc = _lib.lib().vmf_elbo_bwd(None, None, _lib.ptr(kap), _lib.ptr(go), None, _lib.ptr(dk2), a_rec, a_kl, n_x, n_hat, 777,
                                 _lib.KL_SPHERICAL_PIECEWISE, _lib.VMF_F32, _lib.stream_ptr(DEV))
    assert rc == 0
    assert torch.equal(dk, dk2)


@pytest.mark.parametrize("shape_x,shape_h", [((5, 7), (5, 7)), ((5, 7), (3, 5, 7)), ((1, 7), (5, 7)), ((5, 1), (5, 6)),
                                             ((6,), (2, 3, 6)), ((4, 8), (4, 4, 8)), ((2, 3), (5, 2, 3))])
@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_elbo_broadcast_shapes_vs_torch_autograd(shape_x, shape_h, tag):
    """scalar path, 128-bit path, materialised broadcasts, gradient of x"""
    gen = torch.Generator(device=DEV).manual_seed(5)
    dt = TD[tag]
    x = torch.randn(*shape_x, device=DEV, dtype=dt, generator=gen).requires_grad_()
    xh = torch.randn(*shape_h, device=DEV, dtype=dt, generator=gen).requires_grad_()
    kap = (torch.rand(9, device=DEV, dtype=dt, generator=gen) * 30 + 0.5).requires_grad_()
    up = torch.tensor(1.7, device=DEV, dtype=dt)
    (losses.elbo_loss(x, xh, kap, 0.9, 0.4, "torus") * up).backward()
    got = [N(t.grad) for t in (x, xh, kap)]
    for t in (x, xh, kap):
        t.grad = None
    sh = torch.sinh(kap)
    kl = torch.log(kap) - torch.log(sh) + kap * (torch.cosh(kap) / sh) - 1.0
    ref = 0.9 * torch.mean((x - xh) * (x - xh)) + 0.4 * kl.mean()
    np.testing.assert_allclose(N(losses.elbo_loss(x, xh, kap, 0.9, 0.4, "torus")), N(ref), **TOL[tag])
    (ref * up).backward()
    for a, t in zip(got, (x, xh, kap)):
        np.testing.assert_allclose(a, N(t.grad), rtol=TOL[tag]["rtol"], atol=10 * TOL[tag]["atol"])


def test_elbo_honours_a_subclass_that_overrides_the_kl_method():
    """model/spherical_vae.py:132-134 calls self.kl_vmf_spherical_uniform: a subclass with its own KL must see it used
    (the fused reduction serves the stock method only), and the stock class must give the reference's composition"""
    from medicalimageanalysis_amd.model.spherical_vae import SphericalVAE
    cfg = dict(extrinsic_dim=12, latent_dim=3, sftbeta=1.0, encoder_width=16, encoder_depth=1, decoder_width=16,
               decoder_depth=1, dropout_p=0.0, device="cuda", gamma=0.7, beta=0.3)

    class Mine(SphericalVAE):
        def kl_vmf_spherical_uniform(self, kappa):
            return 2.0 * kappa   # anything recognisable

    torch.manual_seed(0)
    x = torch.rand(8, 12, device=DEV)
    x_mu = torch.rand(8, 8, 12, device=DEV)
    z_mu = torch.randn(8, 3, device=DEV)
    kap = torch.rand(8, 1, device=DEV) * 30 + 1
    stock, mine = SphericalVAE(cfg), Mine(cfg)
    rec = torch.mean((x - x_mu) ** 2)
    got = mine._elbo(x, x_mu, (z_mu, kap))
    np.testing.assert_allclose(N(got), N(0.7 * rec + 0.3 * (2.0 * kap).mean()), rtol=1e-6)
    got = stock._elbo(x, x_mu, (z_mu, kap))
    np.testing.assert_allclose(N(got), N(0.7 * rec + 0.3 * stock.kl_vmf_spherical_uniform(kap).mean()), rtol=1e-5)


def test_elbo_sum_reduction_is_the_s1_loss():
    """regular_cnn/s1_vae.py:70-92: mse_sum + sum_i (k I1/I0 - log I0)"""
    from medicalimageanalysis_amd.model.s1_vae import loss_function
    gen = torch.Generator(device=DEV).manual_seed(1)
    x = torch.rand(16, 3, 8, 8, device=DEV, generator=gen)
    rx = torch.rand(16, 3, 8, 8, device=DEV, generator=gen).requires_grad_()
    kap = (torch.rand(16, 1, device=DEV, generator=gen) * 20 + 1).requires_grad_()
    val = loss_function(rx, x, None, kap)
    k = kap.squeeze()
    ref = torch.nn.functional.mse_loss(rx, x, reduction="sum") + torch.sum(
        k * (torch.special.i1(k) / torch.special.i0(k)) - torch.log(torch.special.i0(k)))
    np.testing.assert_allclose(N(val), N(ref), rtol=2e-6)
    val.backward()
    g1, g2 = N(rx.grad), N(kap.grad)
    rx.grad = kap.grad = None
    ref.backward()
    np.testing.assert_allclose(g1, N(rx.grad), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(g2, N(kap.grad), rtol=2e-4, atol=1e-5)


def test_elbo_full_size_properties():
    """B = 128, 28x28: x_hat (128, 128, 784) = 51 MB.  loss(x, x) has a zero reconstruction term; scaling the
    residual by 2 scales it by 4; nan in x_hat propagates"""
    gen = torch.Generator(device=DEV).manual_seed(3)
    B, D = 128, 784
    x = torch.rand(B, D, device=DEV, generator=gen)
    kap = torch.rand(B, 1, device=DEV, generator=gen) * 99 + 1
    xh = x.expand(B, B, D).contiguous()
    kl_only = losses.elbo_loss(x, xh, kap, 1.0, 1.0, "spherical")
    np.testing.assert_allclose(N(kl_only), co.elbo(N(x), N(x), N(kap), 0.0, 1.0), rtol=1e-6)
    r = torch.randn(B, B, D, device=DEV, generator=gen)
    a = losses.elbo_loss(x, xh + r, kap, 1.0, 0.0, "spherical")
    b = losses.elbo_loss(x, xh + 2 * r, kap, 1.0, 0.0, "spherical")
    np.testing.assert_allclose(N(b), 4 * N(a), rtol=1e-5)
    np.testing.assert_allclose(N(a), float((r.double() ** 2).mean()), rtol=1e-5)
    xh[5, 7, 11] = float("nan")
    assert torch.isnan(losses.elbo_loss(x, xh, kap, 1.0, 1.0, "spherical"))


# ---- torus embedding -----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("name", ["bb", "b"])
def test_build_torus_matches_goldens(golden, tag, name):
    g = golden("g_consumers")
    k = f"tor_{name}_{tag}::"
    zt, zp = T(g[k + "zt"]).requires_grad_(), T(g[k + "zp"]).requires_grad_()
    out = losses.build_torus(zt, zp, 1.5, 0.5)
    assert out.shape == g[k + "out"].shape
    np.testing.assert_allclose(N(out), g[k + "out"], **TOL[tag])
    (out * T(g[k + "c"])).sum().backward()
    np.testing.assert_allclose(N(zt.grad), g[k + "g_zt"], **TOL[tag])
    np.testing.assert_allclose(N(zp.grad), g[k + "g_zp"], **TOL[tag])


def test_build_torus_tile_kernel_and_row_kernel_give_the_same_bits():
    """fp32 rows of three components, 16-byte aligned: full tiles of 256 rows take the kernel that moves whole tiles, the
    tail and every other layout the thread-per-row kernel -- the same rows through both (aligned; a view that starts one
    element into its buffer) agree bit for bit, forward and backward, and match the formula in double precision"""
    gen = torch.Generator(device=DEV).manual_seed(4)
    n = 256 * 37 + 101
    zt = torch.nn.functional.normalize(torch.randn(n, 3, device=DEV, generator=gen), dim=-1)
    zp = torch.nn.functional.normalize(torch.randn(n, 3, device=DEV, generator=gen), dim=-1)
    c = torch.randn(n, 3, device=DEV, generator=gen)

    def run(a, b):
        a, b = a.detach().requires_grad_(), b.detach().requires_grad_()
        out = losses.build_torus(a, b, 1.5, 0.5)
        (out * c).sum().backward()
        return out.detach(), a.grad, b.grad

    def off(t):
        buf = torch.empty(t.numel() + 1, device=DEV)
        v = buf[1:].view(t.shape)
        v.copy_(t)
        return v

    got = run(zt, zp)
    assert zt.data_ptr() % 16 == 0
    got_row = run(off(zt), off(zp))
    for x, y in zip(got, got_row):
        assert torch.equal(x, y)
    ct, st, cp, sp = zt[:, 0].double(), zt[:, 1].double(), zp[:, 0].double(), zp[:, 1].double()
    ref = torch.stack([(1.5 + 0.5 * ct) * cp, (0.5 + 0.5 * ct) * sp, 0.5 * st], dim=-1)
    np.testing.assert_allclose(N(got[0]), N(ref), rtol=1e-6, atol=1e-6)
    assert float(got[1][:, 2].abs().max()) == 0.0 and float(got[2][:, 2].abs().max()) == 0.0


def test_build_torus_model_method_and_broadcast():
    cfg = dict(extrinsic_dim=6, latent_dim=3, sftbeta=1.0, encoder_width=8, encoder_depth=1, decoder_width=8,
               decoder_depth=1, dropout_p=0.0, device="cuda", gamma=1.0, beta=0.1, longest_radius=1.5,
               shortest_radius=0.5)
    model = ToroidalVAE(cfg)
    gen = torch.Generator(device=DEV).manual_seed(2)
    zt = torch.randn(4, 1, 3, device=DEV, generator=gen).requires_grad_()
    zp = torch.randn(1, 5, 3, device=DEV, generator=gen).requires_grad_()
    out = model._build_torus(zt, zp)
    assert out.shape == (4, 5, 3)
    ref = torch.stack([(1.5 + 0.5 * zt[..., 0]) * zp[..., 0], (0.5 + 0.5 * zt[..., 0]) * zp[..., 1],
                       0.5 * zt[..., 1].expand(4, 5)], dim=-1)
    np.testing.assert_allclose(N(out), N(ref), rtol=1e-6, atol=1e-7)
    c = torch.randn(4, 5, 3, device=DEV, generator=gen)
    (out * c).sum().backward()
    g1, g2 = N(zt.grad), N(zp.grad)
    zt.grad = zp.grad = None
    (ref * c).sum().backward()
    np.testing.assert_allclose(g1, N(zt.grad), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(g2, N(zp.grad), rtol=1e-5, atol=1e-6)
    with pytest.raises(IndexError):
        losses.build_torus(torch.randn(3, 1, device=DEV), torch.randn(3, 1, device=DEV), 1.5, 0.5)


# ---- latent losses -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("ds", list(DATASETS))
@pytest.mark.parametrize("name", ["bb", "b"])
def test_latent_loss_matches_goldens(golden, tag, ds, name):
    g = golden("g_consumers")
    k = f"ll_{ds}_{name}_{tag}::"
    cfg = {"dataset": DATASETS[ds], "longest_radius": 1.5, "shortest_radius": 0.5}
    z = T(g[k + "z"]).requires_grad_()
    val = losses.latent_loss(T(g[k + "labels"]), z, cfg)
    assert val.shape == ()
    np.testing.assert_allclose(N(val), g[k + "loss"], **TOL[tag])
    val.backward()
    gz = g[k + "g_z"]
    np.testing.assert_allclose(N(z.grad), gz, rtol=20 * TOL[tag]["rtol"],
                               atol=(1e-12 if tag == "f64" else 2e-5 * np.abs(gz).max()))


@pytest.mark.parametrize("ds", list(DATASETS))
def test_latent_loss_large_batch_vs_oracle_and_per_row(ds):
    """2^20 rows against the numpy oracle; per-row output of the C entry point; row -> label pairing by modulo"""
    gen = torch.Generator(device=DEV).manual_seed(11)
    B, rep = 1 << 14, 64
    z = torch.randn(rep, B, 3, device=DEV, dtype=torch.float64, generator=gen)
    if ds == "T2":
        z = z * 0.3 + torch.tensor([1.2, 0.9, 0.0], device=DEV, dtype=torch.float64)
    labels = torch.rand(B, 2, device=DEV, dtype=torch.float64, generator=gen) * 3.0
    cfg = {"dataset": DATASETS[ds], "longest_radius": 1.5, "shortest_radius": 0.5}
    val = losses.latent_loss(labels, z, cfg)
    rows_ref = co.latent_loss_rows(N(labels), N(z), DATASETS[ds], 1.5, 0.5)
    np.testing.assert_allclose(N(val), rows_ref.mean(), rtol=1e-12)
    n = rep * B
    rows = torch.empty(n, device=DEV, dtype=torch.float64)
    mean = torch.empty((), device=DEV, dtype=torch.float64)
    ws = torch.empty(_lib.KL_WORKSPACE_DOUBLES, device=DEV, dtype=torch.float64)
    rc = _lib.lib().vmf_latent_loss_fwd(_lib.ptr(z), _lib.ptr(labels), _lib.ptr(rows), _lib.ptr(mean), _lib.ptr(ws), n, 3,
                                        B, 2, list(DATASETS).index(ds), 1.5, 0.5, _lib.VMF_F64, _lib.stream_ptr(DEV))
    assert rc == 0
    np.testing.assert_allclose(N(rows).reshape(rep, B), rows_ref, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(N(mean), N(val), rtol=1e-14)


@pytest.mark.parametrize("ds", list(DATASETS))
@pytest.mark.parametrize("n_labels,rep", [(4099, 5), (1024, 4), (3, 333), (1, 1027), (257, 9)])
def test_latent_loss_fp32_layouts_agree_with_fp64(ds, n_labels, rep):
    """fp32 takes four rows per lane when z has three components and is 16-byte aligned, one row per lane otherwise:
    both against the double-precision kernel on the same values -- row counts that are not multiples of four, label
    counts that do (not) divide the quad, a view that starts mid-allocation, per-row output and gradient"""
    gen = torch.Generator(device=DEV).manual_seed(n_labels)
    n = n_labels * rep
    base = torch.randn(n * 3 + 1, device=DEV, dtype=torch.float64, generator=gen)
    labels64 = torch.rand(n_labels, 2, device=DEV, dtype=torch.float64, generator=gen) * 6.0
    variant = list(DATASETS).index(ds)
    lib, sp = _lib.lib(), _lib.stream_ptr(DEV)
    ws = torch.empty(_lib.KL_WORKSPACE_DOUBLES, device=DEV, dtype=torch.float64)

    def run(z, labels):
        dt = _lib.dtype_code(z)
        rows = torch.full((n,), float("nan"), device=DEV, dtype=z.dtype)
        mean = torch.full((), float("nan"), device=DEV, dtype=z.dtype)
        assert lib.vmf_latent_loss_fwd(_lib.ptr(z), _lib.ptr(labels), _lib.ptr(rows), _lib.ptr(mean), _lib.ptr(ws), n, 3,
                                       n_labels, 2, variant, 1.5, 0.5, dt, sp) == 0
        mean2 = torch.full((), float("nan"), device=DEV, dtype=z.dtype)
        assert lib.vmf_latent_loss_fwd(_lib.ptr(z), _lib.ptr(labels), None, _lib.ptr(mean2), _lib.ptr(ws), n, 3,
                                       n_labels, 2, variant, 1.5, 0.5, dt, sp) == 0
        dz = torch.full((n, 3), float("nan"), device=DEV, dtype=z.dtype)
        gm = torch.full((), 2.0, device=DEV, dtype=z.dtype)
        assert lib.vmf_latent_loss_bwd(_lib.ptr(z), _lib.ptr(labels), _lib.ptr(gm), _lib.ptr(dz), n, 3, n_labels, 2,
                                       variant, 1.5, 0.5, dt, sp) == 0
        torch.cuda.synchronize()
        assert N(mean).tobytes() == N(mean2).tobytes()
        return N(rows).astype(np.float64), float(mean), N(dz).astype(np.float64)

    for off in (0, 1):   # off = 1: the fp32 view is 4-byte aligned only -> one row per lane
        z64 = base[off:off + 3 * n].view(n, 3)
        if ds == "T2":
            z64 = z64 * 0.3 + torch.tensor([1.2, 0.9, 0.0], device=DEV, dtype=torch.float64)
        z32 = torch.empty(n * 3 + 4, device=DEV, dtype=torch.float32)[off:off + 3 * n].view(n, 3)
        z32.copy_(z64)
        l32 = labels64.float()
        r64, m64, d64 = run(z32.double().contiguous(), l32.double())
        r32, m32, d32 = run(z32, l32)
        np.testing.assert_allclose(r32, r64, rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(m32, m64, rtol=1e-5)
        # the gradient has poles (rho -> 0, |z_3| -> |z|): compare where it is tame
        scale = np.abs(d64).max(axis=1, keepdims=True)
        ok = (scale < 1e3 / n).repeat(3, axis=1)
        np.testing.assert_allclose(d32[ok], d64[ok], rtol=2e-3, atol=1e-3 / n)
        assert np.isfinite(d32).all() == np.isfinite(d64).all()


def test_latent_loss_errors_and_empty():
    z = torch.randn(4, 3, device=DEV)
    lab = torch.rand(4, 2, device=DEV)
    with pytest.raises(ValueError, match="Unknown dataset"):
        losses.latent_loss(lab, z, {"dataset": "points_dataset"})
    with pytest.raises(RuntimeError):
        losses.latent_loss(torch.rand(3, 2, device=DEV), z, {"dataset": "S2_dataset"})
    with pytest.raises(RuntimeError, match="MI355X only"):
        losses.latent_loss(lab.cpu(), z.cpu(), {"dataset": "S1_dataset"})
    # labels with one column are enough for S1 only
    losses.latent_loss(lab[:, :1].contiguous(), z, {"dataset": "S1_dataset"})
    with pytest.raises(RuntimeError, match="columns"):
        losses.latent_loss(lab[:, :1].contiguous(), z, {"dataset": "S2_dataset"})
    # unknown variant through the C ABI
    ws = torch.empty(_lib.KL_WORKSPACE_DOUBLES, device=DEV, dtype=torch.float64)
    out = torch.empty((), device=DEV)
    rc = _lib.lib().vmf_latent_loss_fwd(_lib.ptr(z), _lib.ptr(lab), None, _lib.ptr(out), _lib.ptr(ws), 4, 3, 4, 2, 7,
                                        0.0, 0.0, _lib.VMF_F32, _lib.stream_ptr(DEV))
    assert rc == -1


def test_train_step_t2_with_fused_consumers_matches_torch_ops():
    """ToroidalVAE step as model/train.py:67-80 runs it (two _elbo calls + latent loss on the embedded means):
    loss and parameter gradients against the same step written with torch ops on the device"""
    cfg = dict(extrinsic_dim=10, latent_dim=3, sftbeta=1.0, encoder_width=16, encoder_depth=2, decoder_width=16,
               decoder_depth=2, dropout_p=0.0, device="cuda", gamma=1.0, beta=0.05, alpha=0.3, longest_radius=1.5,
               shortest_radius=0.5, dataset="T2_dataset")
    torch.manual_seed(0)
    model = ToroidalVAE(cfg).double()
    B = 12
    gen = torch.Generator(device=DEV).manual_seed(9)
    x = torch.rand(B, 10, device=DEV, dtype=torch.float64, generator=gen)
    labels = torch.rand(B, 2, device=DEV, dtype=torch.float64, generator=gen) * 6.28

    def step(fused):
        torch.manual_seed(4)
        for p in model.parameters():
            p.grad = None
        z, x_mu, (tm, tk, pm, pk) = model(x)
        if fused:
            zl = model._build_torus(tm, pm)
            loss = model._elbo(x, x_mu, (tm, tk)) + model._elbo(x, x_mu, (pm, pk))
            loss = loss + cfg["alpha"] * losses.latent_loss(labels, zl, cfg)
        else:
            R, r = 1.5, 0.5
            zl = torch.stack([(R + r * tm[..., 0]) * pm[..., 0], (r + r * tm[..., 0]) * pm[..., 1], r * tm[..., 1]], -1)
            rec = torch.mean((x - x_mu) * (x - x_mu))

            def kl(k):
                return (torch.log(k) - torch.log(torch.sinh(k)) + k * torch.cosh(k) / torch.sinh(k) - 1.0).mean()
            loss = 2 * cfg["gamma"] * rec + cfg["beta"] * (kl(tk) + kl(pk))
            xx, yy, zc = zl[..., 0], zl[..., 1], zl[..., 2]
            rho = torch.sqrt(xx ** 2 + yy ** 2)
            ph = (torch.atan2(yy, xx) + 2 * torch.pi) % (2 * torch.pi)
            th = (torch.atan2(zc, rho - R) + 2 * torch.pi) % (2 * torch.pi)
            ll = torch.mean(1 - torch.cos(th - labels[:, 0])) + torch.mean(1 - torch.cos(ph - labels[:, 1])) + \
                0.1 * ((rho - R) ** 2 + zc ** 2 - r ** 2) ** 2
            loss = loss + cfg["alpha"] * ll.mean()
        loss.backward()
        return float(loss.detach()), [p.grad.clone() for p in model.parameters()]

    la, ga = step(True)
    lb, gb = step(False)
    assert abs(la - lb) < 1e-11 * max(1, abs(lb))
    for a, b in zip(ga, gb):
        np.testing.assert_allclose(N(a), N(b), rtol=1e-8, atol=1e-11)


# ---- multi-rank path on the device (two processes sharing the one GPU of the test box, gloo for the scalars) --------
_RANK_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from medicalimageanalysis_amd import kl as klmod, rng
from medicalimageanalysis_amd.distributed import shard_rows, allreduce_scalars
from medicalimageanalysis_amd.sampler.von_mises_fisher import VonMisesFisher
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda:0")
n, m = 100003, 64
g = torch.Generator(device=dev).manual_seed(0)            # every rank builds the same global inputs
mu = torch.nn.functional.normalize(torch.randn(n, m, device=dev, generator=g), dim=-1)
kappa = torch.exp(torch.rand(n, device=dev, generator=g) * float(np.log(1e4)))
a, b = shard_rows(n, rank, world)
q = VonMisesFisher(mu[a:b].contiguous(), kappa[a:b].contiguous(), validate_args=False, row_offset=a,
                   generator=rng.PhiloxGenerator(99, 5))
z, logp = q.rsample(return_log_prob=True)
full = VonMisesFisher(mu, kappa, validate_args=False, generator=rng.PhiloxGenerator(99, 5))
zf, logpf = full.rsample(return_log_prob=True)
assert torch.equal(z, zf[a:b]) and torch.equal(logp, logpf[a:b])      # sharded draw == unsharded draw, bit for bit
kl_local = klmod.kl_sum(kappa[a:b].contiguous(), "stable_m3").double()
tot = allreduce_scalars(kl_local, float(b - a), logp.double().sum())
kl_full = klmod.kl_sum(kappa, "stable_m3").double()
assert abs(tot[0].item() - kl_full.item()) < 1e-5 * abs(kl_full.item()), (tot[0].item(), kl_full.item())
assert tot[1].item() == n
assert abs(tot[2].item() - logpf.double().sum().item()) < 1e-6 * abs(logpf.double().sum().item())
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_rank_sharded_sampling_on_the_device(tmp_path):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rank_worker.py"
    script.write_text(_RANK_WORKER % {"root": root})
    port = 29500 + os.getpid() % 2000
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   VMF_RNG="philox")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=100)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o


def test_bench_launches_two_ranks_itself_on_the_shared_device():
    """`python bench.py --gpus 2` with no launcher: the parent starts both ranks before touching the GPU; on a one-GPU
    box they share the device (gloo rendezvous).  Both ranks' kernel times come back and the rate counts both."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--rows",
                          str(1 << 20), "--cpu-rows", "0", "--steps", "5", "--warmup", "2", "--with-allreduce"], cwd=root,
                         env=env, capture_output=True, text=True, timeout=110)
    assert run.returncode == 0, run.stdout + run.stderr
    lines = [l for l in run.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, run.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and "rehearsal" not in rec
    assert len(rec["kernel_ms_per_rank"]) == 2 and all(t and t > 0 for t in rec["kernel_ms_per_rank"])
    assert abs(rec["value"] - 2 * (1 << 20) * 5 / (rec["ms_per_step"] * 5e-3)) < 1e-6 * rec["value"]
    assert rec["allreduce"]["count_seen"] == 2 * (1 << 20) and rec["roofline"]["frac"] > 0


@pytest.mark.parametrize("config", [1, 2, 4, 5])
def test_bench_other_configurations_produce_a_record(config):
    """every BASELINE configuration comes out of bench.py as one JSON line with roofline and cpu_baseline (sizes cut
    down here; the driver-sized runs are in profiles/)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows = {1: 4096, 2: 128, 4: 1 << 14, 5: 1 << 12}[config]
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", str(config), "--rows", str(rows),
                          "--cpu-rows", "2048", "--steps", "3", "--warmup", "2"], cwd=root, capture_output=True, text=True,
                         timeout=110)
    assert run.returncode == 0, run.stdout + run.stderr
    rec = json.loads(run.stdout.strip().splitlines()[-1])
    assert rec["config"]["bench_config"] == config and rec["value"] > 0 and rec["n_gpus"] == 1
    assert rec["roofline"]["bound"] == "hbm" and rec["roofline"]["achieved"] > 0
    assert rec["cpu_baseline"]["value"] > 0 and rec["cpu_baseline"]["cores"] >= 1


def test_c_abi_driver_without_python_runs_the_headline_kernel():
    """medicalimageanalysis_amd/abi_bench (tools/abi_bench.hip) links only libvmf_hip.so and the HIP runtime: the
    same entry points from plain C++ with hipMalloc'ed buffers"""
    import json
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(_lib.library_path()), "abi_bench")
    if not os.path.exists(exe):
        pytest.skip("abi_bench not built")
    run = subprocess.run([exe, str(1 << 20), "64", "5", "2"], capture_output=True, text=True, timeout=100)
    assert run.returncode == 0, run.stdout + run.stderr
    out = json.loads(run.stdout.strip().splitlines()[-1])
    assert out["self_check"] == "ok" and out["rows"] == 1 << 20 and out["kernel_ms_min"] > 0
    run = subprocess.run([exe, "1000", "10", "2", "1"], capture_output=True, text=True, timeout=100)   # scalar-access shape
    assert run.returncode == 0, run.stdout + run.stderr


def test_allreduce_scalars_through_a_one_rank_rccl_communicator():
    """vmf_allreduce_scalars against a communicator made with RCCL's own C API (the library torch ships): with one
    rank the sum leaves the packed buffer unchanged; the call is asynchronous on the given stream."""
    import os
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    torch.cuda.set_device(DEV)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        vals = [1.5, 2.5, -3.0, 128.0]
        buf = torch.tensor(vals, device=DEV, dtype=torch.float64)
        rc = _lib.lib().vmf_allreduce_scalars(comm, _lib.ptr(buf), 4, _lib.stream_ptr(DEV))
        assert rc == 0, _lib.lib().vmf_last_error_string()
        torch.cuda.synchronize()
        assert buf.tolist() == vals
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
