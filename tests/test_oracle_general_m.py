"""General-m sampler pieces of the CPU oracle (Wood rejection, symmetric Beta, Householder, Philox,
log-normaliser, entropy, KL).  The reference has no general-m code (it asserts m == 3,
sampler/von_mises_fisher.py:28), so these are pinned by known-answer vectors, closed forms, scipy and
distribution tests: parity for them is unpinned by the reference.
"""
import numpy as np
import pytest
import scipy.integrate
import scipy.special as sp
import scipy.stats as st


def test_philox_known_answers(oracle):
    """Random123 kat_vectors for philox4x32-10"""
    assert oracle.philox_raw([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox_raw([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox_raw([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == [
        0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_u01_open_interval_and_fp32_exact(oracle):
    lib = oracle.lib()
    for x in (0, 1, 511, 512, 0xffffffff, 0x80000000):
        u = lib.orc_u01(x)
        assert 0.0 < u < 1.0
        assert float(np.float32(u)) == u


def vmf_w_cdf(m, kappa):
    """CDF of the cosine marginal  p(w) ~ exp(kappa w) (1-w^2)^((m-3)/2)  on [-1,1] (log-domain grid)"""
    t = np.linspace(0, np.pi, 200001)
    w = np.cos(t)[::-1]
    # integrate in theta to avoid endpoint singularities: dw = sin(t) dt
    th = t[::-1]
    dens = np.exp(kappa * (np.cos(th) - 1) + (m - 2) * np.log(np.maximum(np.sin(th), 1e-300)))
    cdf = scipy.integrate.cumulative_trapezoid(dens, -th, initial=0)
    cdf /= cdf[-1]
    return lambda x: np.interp(x, w, cdf)


@pytest.mark.parametrize("m,kappa", [(3, 5.0), (4, 0.5), (10, 30.0), (64, 1.0), (64, 100.0), (64, 1e4), (512, 1e3),
                                     (2, 3.0)])
def test_wood_w_distribution(oracle, m, kappa):
    n = 20000
    mu = np.zeros((n, m))
    mu[:, 0] = 1.0
    z, w, idx = oracle.vmf_rsample_philox(mu, np.full(n, kappa), seed=1234 + m)
    assert np.all(idx < 63)
    p = st.kstest(w, vmf_w_cdf(m, kappa)).pvalue
    assert p > 1e-3, (m, kappa, p)
    # mu = e1 -> Householder is the identity -> z[:,0] == w ; rows are unit vectors
    np.testing.assert_allclose(z[:, 0], w, atol=1e-12)
    np.testing.assert_allclose(np.linalg.norm(z, axis=1), 1.0, atol=1e-9)
    # E[w] = A_m(kappa)
    A = sp.ive(m / 2, kappa) / sp.ive(m / 2 - 1, kappa)
    assert abs(w.mean() - A) < 6 * w.std() / np.sqrt(n)


def test_m3_wood_agrees_with_inverse_cdf(oracle):
    """for m = 3 the rejection sampler and the reference's inverse CDF share one law"""
    n, kappa = 20000, 7.0
    mu = np.zeros((n, 3)); mu[:, 2] = 1.0
    _, w, _ = oracle.vmf_rsample_philox(mu, np.full(n, kappa), seed=5)
    u = (np.arange(n) + 0.5) / n
    w3 = 1 + np.log(u + (1 - u) * np.exp(-2 * kappa)) / kappa
    assert st.ks_2samp(w, w3).pvalue > 1e-3


def test_householder_moves_pole_to_mu(oracle):
    rng = np.random.default_rng(0)
    n, m = 4000, 16
    mu = rng.standard_normal((n, m)); mu /= np.linalg.norm(mu, axis=1, keepdims=True)
    kappa = np.full(n, 50.0)
    z, w, _ = oracle.vmf_rsample_philox(mu, kappa, seed=9)
    np.testing.assert_allclose(np.linalg.norm(z, axis=1), 1.0, atol=1e-9)
    np.testing.assert_allclose((z * mu).sum(1), w, atol=1e-9)  # reflection is an isometry: <z,mu> = <z',e1> = w
    # mean direction
    zbar = z.mean(0)
    assert np.linalg.norm(zbar) < 0.2  # mu are isotropic


def test_pre_drawn_equals_philox_path(oracle):
    """feeding the proposals/gaussians the counter-based path would draw reproduces it exactly"""
    lib = oracle.lib()
    rng = np.random.default_rng(1)
    n, m, K = 300, 8, 16
    mu = rng.standard_normal((n, m)); mu /= np.linalg.norm(mu, axis=1, keepdims=True)
    kappa = np.exp(rng.uniform(0, np.log(1e3), n))
    seed, off = 77, 3
    z1, w1, i1 = oracle.vmf_rsample_philox(mu, kappa, seed=seed, offset=off, row0=1000, max_prop=K)
    pe = np.empty((n, K)); pu = np.empty((n, K)); ga = np.empty((n, m))
    import ctypes as C
    out = (C.c_uint32 * 4)()
    for r in range(n):
        for k in range(K):
            lib.orc_philox4x32_10(C.c_uint64(1000 + r), C.c_uint32(k), C.c_uint64(seed), C.c_uint64(off), out)
            U, V, ua = (lib.orc_u01(out[j]) for j in range(3))
            s = -np.expm1(2.0 / (m - 2) * np.log(U))
            pe[r, k] = 0.5 + 0.5 * np.sqrt(s) * np.cos(2 * np.pi * V)
            pu[r, k] = ua
        for g in range(m // 4):
            lib.orc_philox4x32_10(C.c_uint64(1000 + r), C.c_uint32(0x10000 + g), C.c_uint64(seed), C.c_uint64(off), out)
            for h in range(2):
                u0, u1 = lib.orc_u01(out[2 * h]), lib.orc_u01(out[2 * h + 1])
                rr = np.sqrt(-2 * np.log(u0))
                ga[r, 4 * g + 2 * h] = rr * np.cos(2 * np.pi * u1)
                ga[r, 4 * g + 2 * h + 1] = rr * np.sin(2 * np.pi * u1)
    z2, w2, i2 = oracle.vmf_rsample_pre(mu, kappa, pe, pu, ga)
    np.testing.assert_array_equal(i1, i2)
    np.testing.assert_allclose(w1, w2, atol=1e-12)
    np.testing.assert_allclose(z1, z2, atol=1e-12)


def test_symmetric_beta_law(oracle):
    """e = (1-w)/((1+b) - w(1-b)) recovered from accepted w at kappa -> 0 is Beta((m-1)/2,(m-1)/2)-like;
    checked directly on the proposal formula instead"""
    rng = np.random.default_rng(3)
    for m in (3, 5, 64, 512):
        U, V = rng.uniform(size=50000), rng.uniform(size=50000)
        e = 0.5 + 0.5 * np.sqrt(-np.expm1(2.0 / (m - 2) * np.log(U))) * np.cos(2 * np.pi * V)
        a = (m - 1) / 2
        assert st.kstest(e, st.beta(a, a).cdf).pvalue > 1e-3, m


def test_log_ive_and_ratio_against_scipy(oracle):
    for v in (0.0, 0.5, 1.0, 4.0, 31.0, 32.0, 255.0, 256.0):
        x = np.logspace(-2, 5, 57)
        ref = sp.ive(v, x)
        ok = ref > 1e-290
        np.testing.assert_allclose(oracle.log_ive(v, x)[ok], np.log(ref[ok]), rtol=0, atol=2e-12 * (1 + v))
        if v >= 1:
            r0 = sp.ive(v - 1, x)
            ok = (ref > 1e-290) & (r0 > 1e-290)
            np.testing.assert_allclose(oracle.bessel_ratio(v, x)[ok], ref[ok] / r0[ok], rtol=1e-10)
    # where scipy underflows the log form stays finite and monotone
    li = oracle.log_ive(255.0, np.array([1e-2, 1.0, 10.0]))
    assert np.all(np.isfinite(li)) and np.all(np.diff(li) > 0)


def test_general_kl_reduces_to_reference_forms(golden, oracle):
    g = golden("g_kl")
    k = g["kappa_f64"]
    sel = (k > 1e-2) & (k < 700)
    _, kl3 = oracle.vmf_entropy_kl(k[sel], 3)
    np.testing.assert_allclose(kl3, g["torus_f64"][sel], rtol=1e-8, atol=1e-9)  # model/torus_vae.py:190-194
    _, kl2 = oracle.vmf_entropy_kl(k[sel], 2)
    np.testing.assert_allclose(kl2, g["s1_f64"][sel], rtol=1e-8, atol=1e-9)     # regular_cnn/s1_vae.py:88


def test_log_prob_integrates_to_one(oracle):
    """Monte-Carlo free check in m = 3: closed form C_3 = k / (4 pi sinh k)"""
    k = np.array([0.1, 1.0, 10.0, 200.0])
    lib = oracle.lib()
    for kk in k:
        ref = np.log(kk) - np.log(4 * np.pi) - (kk + np.log1p(-np.exp(-2 * kk)) - np.log(2))
        assert abs(lib.orc_vmf_log_norm_scalar(3, kk) - ref) < 1e-11
    # m = 2: C_2 = 1 / (2 pi I0)
    for kk in k:
        ref = -np.log(2 * np.pi) - (np.log(sp.i0e(kk)) + kk)
        assert abs(lib.orc_vmf_log_norm_scalar(2, kk) - ref) < 1e-11
    rng = np.random.default_rng(5)
    mu = rng.standard_normal((4, 6)); mu /= np.linalg.norm(mu, axis=1, keepdims=True)
    x = rng.standard_normal((4, 6)); x /= np.linalg.norm(x, axis=1, keepdims=True)
    lp = oracle.vmf_log_prob(mu, k, x)
    v = 2.0
    ref = k * (mu * x).sum(1) + v * np.log(k) - 3 * np.log(2 * np.pi) - (np.log(sp.ive(v, k)) + k)
    np.testing.assert_allclose(lp, ref, rtol=1e-11)
