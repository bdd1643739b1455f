#!/usr/bin/env python3
"""The m = 3 reference-exact kernels through the C ABI, kernel time only (HIP events around the launches, tensors
resident): the row-per-row shape at 2^24 rows and the (B, 1)-scale outer product at B = 4096 and 128, forward and
backward, fp32.  Prints a markdown table; run on the GPU box:  python tools/bench_vmf3.py [--outer-only] [--reps N]
(--reps 3 under rocprofv3 --pmc)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from _timing import timeit as _timeit  # noqa: E402

from medicalimageanalysis_amd import _lib  # noqa: E402
from medicalimageanalysis_amd.sampler.von_mises_fisher import vmf3_broadcast_plan  # noqa: E402

DEV = torch.device("cuda:0")


REPS = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 20


def timeit(fn, reps=None, warm=None):
    reps = reps or REPS
    if warm is None and reps < 10:  # profiler passes: every call is a traced dispatch
        warm = reps
    return _timeit(fn, reps=reps, warm_calls=warm)


def row(name, ms, nbytes):
    print(f"| {name} | {ms:.4f} | {nbytes / ms / 1e6:.0f} | {100 * nbytes / ms / 1e6 / 8000:.1f} % |", flush=True)


def main():
    lib = _lib.lib()
    sp = _lib.stream_ptr(DEV)
    arr = _lib.i64_array
    print("| case (fp32) | kernel ms | algorithmic GB/s | of 8 TB/s |")
    print("|---|---|---|---|")
    g = torch.Generator(device=DEV).manual_seed(0)

    def case(name, loc_shape, scale_shape, given, fwd_bytes, bwd_bytes):
        plan = vmf3_broadcast_plan(loc_shape, scale_shape)
        loc = torch.randn(loc_shape, device=DEV, generator=g)
        kap = torch.rand(scale_shape, device=DEV, generator=g) * 99 + 1
        u = torch.rand(plan["w_shape"], device=DEV, generator=g) if given else None
        eps = torch.randn(plan["e_shape"] + (3,), device=DEV, generator=g) if given else None
        z = torch.empty(plan["o_shape"] + (3,), device=DEV)
        nd = len(plan["odims"])
        pa = [arr(plan[k]) for k in ("odims", "w_str", "k_str", "l_str", "e_str")]

        def f():
            rc = lib.vmf3_rsample_fwd(_lib.ptr(loc), _lib.ptr(kap), _lib.ptr(u), _lib.ptr(eps), _lib.ptr(z), None, None, nd,
                                      *pa, 0, _lib.VMF_F32, 0, 1, 0, None, sp)
            assert rc == 0, lib.vmf_last_error_string()
        row(f"{name} forward, {'given' if given else 'in-kernel'} noise", timeit(f), fwd_bytes)
        gz = torch.randn_like(z)
        dloc, dk = torch.empty_like(loc), torch.empty_like(kap)
        ni, nj = C.c_int64(0), C.c_int64(0)
        kind = lib.vmf3_broadcast_kind(nd, *pa, C.byref(ni), C.byref(nj))
        if kind == _lib.KIND_OUTER:
            nb = C.c_int64(0)
            lib.vmf3_rsample_bwd_outer_workspace(ni.value, nj.value, _lib.VMF_F32, C.byref(nb))
            ws = torch.empty((nb.value + 7) // 8, device=DEV, dtype=torch.float64)

            def b():
                rc = lib.vmf3_rsample_bwd_outer(_lib.ptr(loc), _lib.ptr(kap), _lib.ptr(u), _lib.ptr(eps), _lib.ptr(gz), None,
                                                _lib.ptr(dloc), _lib.ptr(dk), ni.value, nj.value, _lib.ptr(ws),
                                                ws.numel() * 8, 0, _lib.VMF_F32, 0, 1, 0, None, sp)
                assert rc == 0, lib.vmf_last_error_string()
        else:
            def b():
                rc = lib.vmf3_rsample_bwd(_lib.ptr(loc), _lib.ptr(kap), _lib.ptr(u), _lib.ptr(eps), _lib.ptr(gz), None,
                                          _lib.ptr(dloc), _lib.ptr(dk), loc.numel() // 3, kap.numel(), nd, *pa, 0,
                                          _lib.VMF_F32, 0, 1, 0, None, sp)
                assert rc == 0, lib.vmf_last_error_string()
        tb = timeit(b)
        row(f"{name} backward, {'given' if given else 'regenerated'} noise", tb, bwd_bytes)

        def fb():
            f()
            b()
        row(f"{name} forward + backward", timeit(fb), fwd_bytes + bwd_bytes)
        del loc, kap, u, eps, z, gz
        torch.cuda.empty_cache()

    n = 1 << 24
    for given in ((False, True) if "--outer-only" not in sys.argv else ()):
        nz = 16 if given else 0
        case(f"rows, B = 2^24 -> (B, 3)", (n, 3), (n,), given, n * (28 + nz), n * (44 + nz))
    for B in ([int(x) for x in os.environ["VMF3_BENCH_B"].split(",")] if "VMF3_BENCH_B" in os.environ else (4096, 128)):
        for given in (False, True):
            # forward: z written once; backward: grad z read once (the per-row operands are negligible)
            case(f"outer, B = {B} -> ({B}, {B}, 3)", (B, 3), (B, 1), given, B * B * 12, B * B * 12)


if __name__ == "__main__":
    main()
