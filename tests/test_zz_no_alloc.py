"""vmf_kl_fwd_static -- the entry point that allocates nothing (its block partials live in an array of the library's
code object) -- against vmf_kl_fwd with a caller workspace, through the C ABI.  Same kernels and launch geometry, so
the results must agree bit for bit: two calls in a row, the second on a smaller input (what the first call left in the
shared array must not leak into it)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a HIP device", allow_module_level=True)

from medicalimageanalysis_amd import _lib  # noqa: E402

DEV = torch.device("cuda:0")


def _both(kappa, variant, with_out):
    lib = _lib.lib()
    sp = C.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)
    n = kappa.numel()
    alpha = 1.0 / max(n, 1)
    res = []
    for static in (False, True):
        out = torch.full_like(kappa, float("nan")) if with_out else None
        acc = torch.full((1,), float("nan"), device=DEV, dtype=torch.float64)
        if static:
            rc = lib.vmf_kl_fwd_static(_lib.ptr(kappa), _lib.ptr(out), _lib.ptr(acc), alpha, n, variant,
                                       _lib.dtype_code(kappa), sp)
        else:
            ws = torch.full((_lib.KL_WORKSPACE_DOUBLES,), float("nan"), device=DEV, dtype=torch.float64)
            rc = lib.vmf_kl_fwd(_lib.ptr(kappa), _lib.ptr(out), _lib.ptr(acc), _lib.ptr(ws), alpha, n, variant,
                                _lib.dtype_code(kappa), sp)
        assert rc == 0, lib.vmf_last_error_string()
        torch.cuda.synchronize(DEV)
        res.append((acc.cpu().numpy().copy(), None if out is None else out.cpu().numpy().copy()))
    return res


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("variant", [_lib.KL_SPHERICAL_PIECEWISE, _lib.KL_S1_I0I1, _lib.KL_STABLE_M3])
def test_static_scratch_kl_matches_workspace_kl_two_calls_in_a_row(dtype, variant):
    g = torch.Generator(device=DEV).manual_seed(3)
    big = torch.exp(torch.rand((1 << 21) + 13, device=DEV, generator=g, dtype=torch.float64) * np.log(80.0)).to(dtype)
    small = big[: 9000 + 3].clone() * 0.5   # fewer blocks than the first call: most of the array keeps old partials
    tiny = big[: 300 + 1].clone() * 0.25    # one block: the kernel writes the sum itself and no partial is stored
    for kappa in (big, small, tiny):
        for with_out in (False, True):
            (s_ws, o_ws), (s_st, o_st) = _both(kappa, variant, with_out)
            assert np.isfinite(s_ws).all()
            assert s_ws.tobytes() == s_st.tobytes(), (s_ws, s_st)
            if with_out:
                assert o_ws.tobytes() == o_st.tobytes()
    # and against the double-precision sum of the per-element values
    for kappa in (small, tiny):
        (s_ws, o_ws), _ = _both(kappa, variant, True)
        np.testing.assert_allclose(s_ws[0], o_ws.astype(np.float64).mean(), rtol=1e-12)


def test_static_scratch_kl_edge_cases():
    lib = _lib.lib()
    sp = C.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)
    acc = torch.full((1,), 7.0, device=DEV, dtype=torch.float64)
    k = torch.ones(8, device=DEV)
    # n = 0: the sum of nothing
    assert lib.vmf_kl_fwd_static(_lib.ptr(k), None, _lib.ptr(acc), 1.0, 0, 0, _lib.VMF_F32, sp) == 0
    torch.cuda.synchronize(DEV)
    assert float(acc) == 0.0
    # beyond the stated maximum: refused before anything is launched
    rc = lib.vmf_kl_fwd_static(_lib.ptr(k), None, _lib.ptr(acc), 1.0, _lib.KL_STATIC_MAX_N + 1, 0, _lib.VMF_F32, sp)
    assert rc == -4 and b"VMF_KL_STATIC_MAX_N" in lib.vmf_last_error_string()
    assert lib.vmf_kl_fwd_static(None, None, _lib.ptr(acc), 1.0, 8, 0, _lib.VMF_F32, sp) == -1
    assert lib.vmf_kl_fwd_static(_lib.ptr(k), None, _lib.ptr(acc), 1.0, 8, 99, _lib.VMF_F32, sp) == -1
    # per-element output only: no scratch is touched
    out = torch.empty_like(k)
    assert lib.vmf_kl_fwd_static(_lib.ptr(k), _lib.ptr(out), None, 1.0, 8, 4, _lib.VMF_F32, sp) == 0
    torch.cuda.synchronize(DEV)
    np.testing.assert_allclose(out.cpu().numpy(), np.log(1.0) - np.log(np.sinh(1.0)) + 1.0 / np.tanh(1.0) - 1.0,
                               rtol=1e-5)


def test_static_scratch_kl_allocates_nothing_and_replays_under_graph_capture():
    """what the entry point is for: after one warm call (which resolves the array's address) the call allocates no
    device memory -- neither through torch's allocator nor behind it -- and can be captured into a HIP graph on a side
    stream; replays on changed concentrations give the bits of vmf_kl_fwd"""
    lib = _lib.lib()
    g = torch.Generator(device=DEV).manual_seed(9)
    n = (1 << 20) + 5
    kappa = torch.exp(torch.rand(n, device=DEV, generator=g) * np.log(50.0))
    out = torch.empty_like(kappa)
    acc = torch.zeros(1, device=DEV, dtype=torch.float64)
    variant = _lib.KL_STABLE_M2

    def call(stream):
        rc = lib.vmf_kl_fwd_static(_lib.ptr(kappa), _lib.ptr(out), _lib.ptr(acc), 1.0 / n, n, variant, _lib.VMF_F32,
                                   C.c_void_p(stream.cuda_stream))
        assert rc == 0, lib.vmf_last_error_string()

    call(torch.cuda.current_stream(DEV))   # warm call: the address look-up happens here
    torch.cuda.synchronize(DEV)
    free0, _ = torch.cuda.mem_get_info(DEV)
    alloc0, reserved0 = torch.cuda.memory_allocated(DEV), torch.cuda.memory_reserved(DEV)
    for _ in range(3):
        call(torch.cuda.current_stream(DEV))
    torch.cuda.synchronize(DEV)
    free1, _ = torch.cuda.mem_get_info(DEV)
    assert torch.cuda.memory_allocated(DEV) == alloc0 and torch.cuda.memory_reserved(DEV) == reserved0
    assert free1 == free0, (free0, free1)
    side = torch.cuda.Stream(DEV)
    graph = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            call(side)
    for scale in (0.5, 1.7):
        kappa.mul_(scale)
        graph.replay()
        torch.cuda.synchronize(DEV)
        got_sum, got_out = acc.cpu().numpy().copy(), out.cpu().numpy().copy()
        (s_ws, o_ws), _ = _both(kappa, variant, True)
        assert got_sum.tobytes() == s_ws.tobytes() and got_out.tobytes() == o_ws.tobytes()
