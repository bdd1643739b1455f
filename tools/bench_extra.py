#!/usr/bin/env python3
"""Secondary measurements on one MI355X (not the headline metric): per-kernel time and algorithmic GB/s for the other
configurations of the path.  Prints a markdown table; run on the GPU box:  python tools/bench_extra.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from _timing import timeit as _timeit  # noqa: E402

from medicalimageanalysis_amd import _lib, kl as klmod, rng  # noqa: E402
from medicalimageanalysis_amd.sampler.ive import ive  # noqa: E402
from medicalimageanalysis_amd.sampler.von_mises_fisher import VonMisesFisher, VonMisesFisher3D  # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, reps=30, warm=None):
    return _timeit(fn, reps=reps, warm_calls=warm)


def row(name, ms, nbytes, units, unit_name):
    print(f"| {name} | {ms:.3f} | {nbytes / ms / 1e6:.0f} | {units / ms / 1e6:.2f} G {unit_name}/s |")


def main():
    lib = _lib.lib()
    sp = _lib.stream_ptr(DEV)
    print("| case | ms | algorithmic GB/s | rate |")
    print("|---|---|---|---|")
    g = torch.Generator(device=DEV).manual_seed(0)

    # --- general m forward: m = 64 with fused log_prob, m = 512, m = 10 (S^9), fp64 m = 64
    for m, n, dt, logp in ((64, 1 << 24, torch.float32, True), (512, 1 << 22, torch.float32, False),
                           (10, 1 << 24, torch.float32, False), (64, 1 << 22, torch.float64, False),
                           (2, 1 << 24, torch.float32, False)):
        mu = torch.nn.functional.normalize(torch.randn(n, m, device=DEV, generator=g, dtype=torch.float32), dim=-1).to(dt)
        # kappa log-uniform on [1, 1e4]; the m = 512 configuration goes up to 1e5
        kap = torch.exp(torch.rand(n, device=DEV, generator=g) * np.log(1e5 if m == 512 else 1e4)).to(dt)
        z = torch.empty_like(mu)
        lp = torch.empty(n, device=DEV, dtype=dt) if logp else None
        es = 4 if dt == torch.float32 else 8

        def f():
            rc = lib.vmf_rsample_fwd(_lib.ptr(mu), m, _lib.ptr(kap), 1, _lib.ptr(z), None, None, None, _lib.ptr(lp), None, n, m,
                                     None, None, 0, None, 64, 0, 1, 0, None, 0, _lib.dtype_code(mu), sp)
            assert rc == 0
        ms = timeit(f)
        row(f"rsample fwd m={m} n=2^{int(np.log2(n))} {str(dt)[6:]}" + (" + fused log_prob" if logp else ""), ms,
            n * (2 * m + 1 + (1 if logp else 0)) * es, n, "rows")
        if m == 64 and dt == torch.float32:
            w = torch.empty(n, device=DEV)
            omw = torch.empty(n, device=DEV)
            lib.vmf_rsample_fwd(_lib.ptr(mu), m, _lib.ptr(kap), 1, _lib.ptr(z), _lib.ptr(w), _lib.ptr(omw), None, None, None, n, m,
                                None, None, 0, None, 64, 0, 1, 0, None, 0, _lib.VMF_F32, sp)
            gz = torch.randn_like(mu)
            dmu = torch.empty_like(mu)
            dk = torch.empty(n, device=DEV)

            def fb():
                rc = lib.vmf_rsample_bwd(_lib.ptr(mu), m, _lib.ptr(kap), 1, _lib.ptr(w), _lib.ptr(omw), _lib.ptr(gz), None,
                                         None, _lib.ptr(dmu), _lib.ptr(dk), n, m, 0, 1, 0, None, 0, _lib.VMF_F32, sp)
                assert rc == 0
            ms = timeit(fb)
            row("rsample bwd m=64 n=2^24 float32 (noise regenerated)", ms, n * (3 * m + 3) * 4, n, "rows")
            q = VonMisesFisher(mu, kap, validate_args=False)
            ms = timeit(lambda: q.log_prob(z))
            row("log_prob m=64 n=2^24 float32", ms, n * (2 * m + 2) * 4, n, "rows")
            del gz, dmu
        del mu, z
        torch.cuda.empty_cache()

    # --- shared-parameter layout: one (mu, kappa), sample_shape = (N,): only z moves (m * 4 bytes per draw)
    for m, n in ((64, 1 << 24), (512, 1 << 22)):
        mu1 = torch.nn.functional.normalize(torch.randn(1, m, device=DEV, generator=g), dim=-1)
        k1 = torch.tensor([50.0], device=DEV)
        z = torch.empty(n, m, device=DEV)

        def fs():
            rc = lib.vmf_rsample_fwd(_lib.ptr(mu1), 0, _lib.ptr(k1), 0, _lib.ptr(z), None, None, None, None, None, n, m, None, None,
                                     0, None, 64, 0, 1, 0, None, 0, _lib.VMF_F32, sp)
            assert rc == 0
        ms = timeit(fs)
        row(f"rsample fwd, one (mu, kappa) shared by all rows, m={m} n=2^{int(np.log2(n))} float32", ms, n * m * 4, n, "rows")
        del z
    torch.cuda.empty_cache()

    # --- m = 3 reference-exact path
    for n in (4096, 1 << 24):
        loc = torch.randn(n, 3, device=DEV, generator=g)
        kap = torch.rand(n, device=DEV, generator=g) * 99 + 1
        q = VonMisesFisher3D(loc, kap, validate_args=False)
        os.environ["VMF_RNG"] = "philox"
        ms = timeit(lambda: q.rsample())
        row(f"VonMisesFisher3D.rsample B={n} fp32, in-kernel noise (incl. python)", ms, n * 28, n, "rows")
        os.environ["VMF_RNG"] = "torch"
        ms = timeit(lambda: q.rsample())
        row(f"VonMisesFisher3D.rsample B={n} fp32, torch noise in reference order", ms, n * 44, n, "rows")
        ms = timeit(lambda: klmod.kl_mean(kap, "spherical"))
        row(f"KL mean (spherical piecewise) n={n} fp32", ms, n * 4, n, "elems")
    for B in (128, 4096):  # the (B,1)-scale broadcast of the reference's VAE: the result is (B,B,3)
        loc = torch.randn(B, 3, device=DEV, generator=g)
        kap = torch.rand(B, 1, device=DEV, generator=g) * 99 + 1
        q = VonMisesFisher3D(loc, kap, validate_args=False)
        for mode in ("philox", "torch"):
            os.environ["VMF_RNG"] = mode
            ms = timeit(lambda: q.rsample())
            # the (B, B, 3) result is written once (12 B per output row); loc, scale -- and u, eps with torch noise -- are
            # O(B) inputs
            row(f"VonMisesFisher3D.rsample B={B} scale (B,1) -> ({B},{B},3), {mode} noise (incl. python)", ms,
                B * B * 12 + B * (16 if mode == "philox" else 32), B * B, "rows")
        if B == 4096:
            zq = q.rsample().detach().requires_grad_(False)
            lq = loc.clone().requires_grad_()
            kq = kap.clone().requires_grad_()
            gzq = torch.randn_like(zq)

            def fbq():
                lq.grad = kq.grad = None
                VonMisesFisher3D(lq, kq, validate_args=False).rsample().backward(gzq)
            ms = timeit(fbq)
            # forward writes z, backward reads grad z: 12 B per output row each; operands, gradients and the backward's
            # partial sums are O(B) and O(B * B / tile height)
            row(f"  forward + backward of the same (B={B})", ms, B * B * (12 + 12) + B * 64, B * B, "rows")
            del zq, gzq
    os.environ["VMF_RNG"] = "torch"

    # --- Spherical-VAE train step, batch 128, 28x28 inputs, latent m = 3: this package vs the same step written with
    #     torch ops following the reference's formulas (what the reference executes on a GPU)
    from medicalimageanalysis_amd.model.spherical_vae import SphericalVAE
    cfg = dict(extrinsic_dim=784, latent_dim=3, sftbeta=1.0, encoder_width=128, encoder_depth=4, decoder_width=128,
               decoder_depth=4, dropout_p=0.1, device="cuda", gamma=1.0, beta=0.01)
    torch.manual_seed(0)
    model = SphericalVAE(cfg)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    x = torch.rand(128, 784, device=DEV)
    torch.distributions.Distribution.set_default_validate_args(False)

    def step_ours():
        z, x_mu, params = model(x)
        loss = model._elbo(x, x_mu, params)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    def step_torch():
        z_mu, z_kappa = model.encode(x)
        loc = z_mu / (z_mu.norm(dim=-1, keepdim=True) + 1e-5)
        u = torch.rand(z_kappa.shape, device=DEV)
        w = 1 + torch.logsumexp(torch.stack([u.log(), (1 - u).log() - 2 * z_kappa], 0), 0) / z_kappa
        eps = torch.randn(z_mu.shape, device=DEV)
        v = eps - (eps * loc).sum(-1, keepdim=True) * loc
        v = v / (v.norm(dim=-1, keepdim=True) + 1e-8)
        we = w.unsqueeze(-1)
        z = we * loc + torch.sqrt(torch.clamp(1 - we ** 2, 1e-10, 1.0)) * v
        x_mu = model.decode(z)
        loc2 = z_mu / (z_mu.norm(dim=-1, keepdim=True) + 1e-5)  # the second construction in _elbo  # noqa: F841
        k = torch.clamp(z_kappa, min=1e-8)
        small, large = k < 1e-3, k > 20.0
        mid = ~(small | large)
        kl = torch.zeros_like(k)
        if mid.any():
            km = k[mid]
            sh = torch.sinh(km) + 1e-8
            kl[mid] = torch.log(km) - torch.log(sh) + km * (torch.cosh(km) / sh) - 1.0
        if large.any():
            kL = k[large]
            kl[large] = torch.log(kL) - (kL - np.log(2.0)) + kL * (1.0 + 2.0 * torch.exp(-2.0 * kL)) - 1.0
        loss = cfg["gamma"] * torch.mean((x - x_mu) * (x - x_mu)) + cfg["beta"] * kl.mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    os.environ["VMF_RNG"] = "torch"
    ms_a = timeit(step_ours, reps=30)
    ms_b = timeit(step_torch, reps=30)
    print(f"| SphericalVAE train step B=128 (z 128x128x3): this package | {ms_a:.3f} | - | {1e3 / ms_a:.0f} steps/s |")
    print(f"| same step with the sampler/KL as torch ops (reference formulas) | {ms_b:.3f} | - | {1e3 / ms_b:.0f} steps/s |")

    # --- the same step captured as one HIP graph (forward + ELBO + backward + Adam = one launch)
    from medicalimageanalysis_amd.graph import GraphedTrainStep
    torch.manual_seed(0)
    model_g = SphericalVAE(cfg)
    opt_g = torch.optim.Adam(model_g.parameters(), lr=1e-3, capturable=True, fused=True)

    def loss_fn(mdl, xb):
        z, x_mu, params = mdl(xb)
        return mdl._elbo(xb, x_mu, params)
    for mode in ("torch", "philox"):
        os.environ["VMF_RNG"] = mode
        gstep = GraphedTrainStep(model_g, opt_g, loss_fn, [x], tune_gemms=True)
        ms_g = timeit(lambda: gstep(x), reps=30)
        print(f"| same step as one HIP graph, fused Adam, tuned GEMMs ({mode} noise) | {ms_g:.3f} | - | {1e3 / ms_g:.0f} steps/s |")
    os.environ["VMF_RNG"] = "torch"

    # --- the 28x28 / batch-128 step with the latent on S^9 (m = 10; one sample per row, the general-m sampler)
    from medicalimageanalysis_amd.model.hyperspherical_vae import HypersphericalVAE
    cfg9 = dict(cfg, latent_dim=10)
    torch.manual_seed(0)
    model9 = HypersphericalVAE(cfg9)
    opt9 = torch.optim.Adam(model9.parameters(), lr=1e-3, capturable=True, fused=True)

    def step9():
        z, x_mu, params = model9(x)
        loss = model9._elbo(x, x_mu, params)
        opt9.zero_grad(set_to_none=True)
        loss.backward()
        opt9.step()
    ms_9 = timeit(step9, reps=30)
    print(f"| HypersphericalVAE train step B=128, latent S^9 (z 128x10) | {ms_9:.3f} | - | {1e3 / ms_9:.0f} steps/s |")
    g9 = GraphedTrainStep(model9, opt9, loss_fn, [x], tune_gemms=True)
    ms_9g = timeit(lambda: g9(x), reps=30)
    print(f"| same S^9 step as one HIP graph, fused Adam, tuned GEMMs | {ms_9g:.3f} | - | {1e3 / ms_9g:.0f} steps/s |")

    # --- consumers: fused ELBO on the (128, 128, 784) decoder output, torus embedding, latent losses
    from medicalimageanalysis_amd import losses
    B, D = 128, 784
    xh = torch.rand(B, B, D, device=DEV, requires_grad=True)
    kp = (torch.rand(B, 1, device=DEV) * 50 + 1).requires_grad_()
    ms = timeit(lambda: losses.elbo_loss(x, xh, kp, 1.0, 0.01, "spherical"))
    row("fused ELBO fwd (128,128,784) fp32 (incl. python)", ms, xh.numel() * 4, xh.numel(), "elems")
    ms_t = timeit(lambda: torch.mean((x - xh) * (x - xh)) + 0.01 * klmod.kl_mean(kp, "spherical"))
    row("  same with torch ops for the MSE", ms_t, xh.numel() * 4, xh.numel(), "elems")
    lo = losses.elbo_loss(x, xh, kp, 1.0, 0.01, "spherical")

    def eb():
        xh.grad = None
        lo.backward(retain_graph=True)
    ms = timeit(eb)
    row("fused ELBO bwd (128,128,784) fp32 (incl. python)", ms, xh.numel() * 8, xh.numel(), "elems")
    n = 1 << 24
    zt = torch.nn.functional.normalize(torch.randn(n, 3, device=DEV, generator=g), dim=-1)
    zp = torch.nn.functional.normalize(torch.randn(n, 3, device=DEV, generator=g), dim=-1)
    ms = timeit(lambda: losses.build_torus(zt, zp, 1.5, 0.5))
    row("build_torus n=2^24 fp32", ms, n * 36, n, "rows")
    labels = torch.rand(4096, 2, device=DEV) * 3
    for ds in ("S1_dataset", "S2_dataset", "T2_dataset"):
        c = {"dataset": ds, "longest_radius": 1.5, "shortest_radius": 0.5}
        ms = timeit(lambda: losses.latent_loss(labels, zt.view(-1, 4096, 3), c))
        row(f"latent_loss {ds} n=2^24 fp32", ms, n * 12, n, "rows")
    del zt, zp, xh
    torch.cuda.empty_cache()

    # --- product of circles T^16: 2^23 rows x 16 circles = 2^27 vMF(m=2) draws; S^1 projected normal + I1/I0 KL
    n = 1 << 27
    mu = torch.nn.functional.normalize(torch.randn(n, 2, device=DEV, generator=g), dim=-1)
    kap = torch.exp(torch.rand(n, device=DEV, generator=g) * np.log(1e3))
    z = torch.empty_like(mu)

    def f5():
        rc = lib.vmf_rsample_fwd(_lib.ptr(mu), 2, _lib.ptr(kap), 1, _lib.ptr(z), None, None, None, None, None, n, 2, None, None,
                                 0, None, 64, 0, 1, 0, None, 0, _lib.VMF_F32, sp)
        assert rc == 0
    ms = timeit(f5, reps=10)
    row("T^16: vMF m=2, 2^23 x 16 circles fp32", ms, n * 20, n, "circle draws")

    def f6():
        rc = lib.vmf_s1_projnormal_fwd(_lib.ptr(mu), _lib.ptr(kap), None, _lib.ptr(z), n, 2, 0, 1, 0, None, _lib.VMF_F32, sp)
        assert rc == 0
    ms = timeit(f6, reps=10)
    row("S^1 projected normal 2^27 rows fp32", ms, n * 20, n, "rows")
    ms = timeit(lambda: klmod.kl_sum(kap, "s1"), reps=10)
    row("KL sum (S^1, I1/I0) 2^27 fp32", ms, n * 4, n, "elems")
    del mu, kap, z
    torch.cuda.empty_cache()

    # --- ive
    n = 1 << 24
    x = torch.exp(torch.rand(n, device=DEV, generator=g, dtype=torch.float64) * np.log(1e5))
    for v in (0.5, 31.0, 255.0):
        ms = timeit(lambda: ive(v, x), reps=10)
        row(f"ive({v}, z) n=2^24 fp64 forward", ms, n * 16, n, "evals")
    xr = x.clone().requires_grad_()
    y = ive(31.0, xr)

    def fbw():
        xr.grad = None
        y.backward(torch.ones_like(y), retain_graph=True)
    ms = timeit(fbw, reps=10)
    row("ive(31, z) n=2^24 fp64 backward", ms, n * 24, n, "evals")


if __name__ == "__main__":
    main()
