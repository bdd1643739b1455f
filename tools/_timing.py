"""Kernel timing shared by the bench tools: HIP events around each call, the median of `reps` calls taken once the GPU
has settled.  After an idle period this GPU runs its first tens of milliseconds of work 5-15 % slower (clocks under a
power cap), and the calls timed here last 10 us to 2 ms, so a fixed count of warm-up calls means a different state for
every size: the warm-up is a stretch of TIME instead."""
import time

import numpy as np
import torch


def timeit(fn, reps=30, warm_ms=60.0, min_warm=10, warm_calls=None):
    """warm_calls: a fixed number of warm-up calls instead (profiler passes, where every call is a traced dispatch)"""
    n, t0 = 0, time.perf_counter()
    if warm_calls is not None:
        for _ in range(warm_calls):
            fn()
        n, warm_ms, min_warm = warm_calls, 0.0, 0
    while n < min_warm or (time.perf_counter() - t0) * 1e3 < warm_ms:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        n += 10
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
