#!/usr/bin/env python3
"""Benchmark of the vMF latent-sampling hot path on MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (the configuration the headline metric is quoted on): batch vMF rsample + log_prob, m = 64, fp32,
2^24 rows PER GPU, one (mu_i, kappa_i) per row, mu = normalize(randn), kappa = exp(U[0, ln 1e4]), noise from the
in-kernel Philox streams keyed by the GLOBAL row index (rank r owns rows [r 2^24, (r+1) 2^24)): weak scaling, no
data-path collective.  A step is one pass of `vmf_rsample_fwd` (C ABI of libvmf_hip.so) over the rank's rows with
all inputs resident in HBM; it writes the sample z and, fused in the same kernel, log p(z).  `--no-log-prob` times
the sampler alone.  Rank 0 prints one JSON line.

roofline : algorithmic bytes per row (2m+2)*4 = 520 B (read mu, read kappa, write z, write log p; 516 B without
           log_prob) x rows per launch / the kernel's average duration measured with HIP events on the launch
           stream, against 8000 GB/s.
cpu_baseline : the CPU oracle (oracle/libvmf_oracle.so, same counters, OpenMP over rows) timed on the host cores
           of rank 0 on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # the GPU needs ~10 launches of this size to settle its clocks (per-dispatch times in profiles/: 2.2, 2.0, 1.9 ...
    # then 1.53-1.56 ms), hence the default warm-up
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=1 << 24, help="rows per GPU")
    ap.add_argument("--m", type=int, default=64)
    ap.add_argument("--log-prob", dest="log_prob", action="store_true", default=True,
                    help="write the fused log p(z) as well (520 B/row; the default)")
    ap.add_argument("--no-log-prob", dest="log_prob", action="store_false", help="sampler alone (516 B/row)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank code path on a box with fewer GPUs than ranks)")
    ap.add_argument("--cpu-rows", type=int, default=1 << 22, help="rows of the CPU-baseline sample (0: skip)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic_latest.json"),
                    help="optional file with {'hbm_bytes_per_launch': ...} from a rocprofv3 --pmc pass")
    return ap.parse_args()


def cpu_baseline(m, rows, seed, log_prob):
    """the oracle on the same workload, bounded sample, host cores of this rank"""
    from oracle import vmf_oracle as orc

    gen = np.random.default_rng(1234)
    mu = gen.standard_normal((rows, m), dtype=np.float32)
    mu /= np.linalg.norm(mu, axis=1, keepdims=True)
    kappa = np.exp(gen.uniform(0, np.log(1e4), rows)).astype(np.float32)
    orc.vmf_rsample_philox(mu[:4096], kappa[:4096], seed=seed)  # warm-up (loads the library, spins up threads)
    t0 = time.perf_counter()
    passes = 0
    while True:  # about 10 s of CPU work: repeated passes over the bounded sample, a new noise offset each pass
        z, _, _ = orc.vmf_rsample_philox(mu, kappa, seed=seed, offset=passes, row0=0, max_prop=64)
        if log_prob:
            orc.vmf_log_prob(mu, kappa, z)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= 10.0 or passes >= 200:
            break
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": rows * passes / dt, "unit": "samples/s", "cores": int(orc.lib().orc_num_threads()),
            "kind": "port", "cpu_model": cpu_model, "host_cpus": os.cpu_count(),
            "sample": f"{passes} passes over {rows} rows of the same workload (m={m}, fp32, oracle's Philox"
                      + (", + log_prob" if log_prob else "") + f"), {dt:.1f} s"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # VMF_BENCH_FORCE_DIST=1 runs the process-group path (RCCL init, barriers, the max-reduce of the time) with a single
    # rank too: the rehearsal of the multi-GPU code path that fits a one-GPU box
    use_dist = world > 1 or os.environ.get("VMF_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if use_dist:
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)
    if a.gpus != world and rank == 0:
        print(f"warning: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    from medicalimageanalysis_amd import _lib

    lib = _lib.lib()
    n, m = a.rows, a.m
    row0 = rank * n
    seed = 0x5EED
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    mu = torch.empty((n, m), device=dev, dtype=torch.float32)
    chunk = 1 << 20
    for s in range(0, n, chunk):  # chunked to keep the temporary small
        e = min(n, s + chunk)
        mu[s:e] = torch.nn.functional.normalize(torch.randn(e - s, m, device=dev, generator=g), dim=-1)
    kappa = torch.exp(torch.rand(n, device=dev, generator=g) * float(np.log(1e4)))
    z = torch.empty_like(mu)
    logp = torch.empty(n, device=dev, dtype=torch.float32) if a.log_prob else None
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)

    def step(i):
        rc = lib.vmf_rsample_fwd(_lib.ptr(mu), m, _lib.ptr(kappa), 1, _lib.ptr(z), None, None, None, _lib.ptr(logp), n,
                                 m, None, None, 0, None, 64, row0, seed, i, None, 0, _lib.VMF_F32, sp)
        if rc:
            _lib.check(rc, "vmf_rsample_fwd")

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    t0 = time.perf_counter()
    for i in range(a.steps):
        evs[i][0].record(stream)
        step(a.warmup + i)
        evs[i][1].record(stream)
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_all = [s.elapsed_time(e) for s, e in evs]
    kern_ms = float(np.mean(kern_all))

    # sanity: the output is a set of unit vectors (cheap, outside the timed region)
    nrm = z[:: max(1, n // 4096)].norm(dim=-1)
    assert torch.isfinite(nrm).all() and float((nrm - 1).abs().max()) < 1e-3, "rsample produced non-unit rows"

    copy_gbps = None
    if rank == 0:
        # context for the roofline fraction: what a plain device copy of the same 2 x n x m x 4 bytes reaches on this
        # GPU (outside the timed region; the output buffer is scratch by now)
        cev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        z.copy_(mu)
        for s_, e_ in cev:
            s_.record(stream)
            z.copy_(mu)
            e_.record(stream)
        torch.cuda.synchronize(dev)
        copy_gbps = 2 * mu.numel() * 4 / (min(s_.elapsed_time(e_) for s_, e_ in cev) * 1e-3) / 1e9
    if rank == 0:
        bytes_per_row = (2 * m + 1 + (1 if a.log_prob else 0)) * 4
        achieved = n * bytes_per_row / (kern_ms * 1e-3) / 1e9
        traffic = None
        try:
            with open(a.traffic_json) as f:
                tj = json.load(f)
            if tj.get("rows") == n and tj.get("m") == m and bool(tj.get("log_prob", False)) == bool(a.log_prob):
                traffic = tj.get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            pass
        out = {
            "metric": "vMF rsample samples/sec (m=64, batch 2^24)" if (m == 64 and n == 1 << 24) else
                      f"vMF rsample samples/sec (m={m}, batch {n})",
            "value": world * n * a.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"batch vMF rsample{' + log_prob (fused)' if a.log_prob else ''}, m={m}, {n} rows per "
                                   f"GPU, one (mu,kappa) per row, kappa log-uniform [1,1e4], in-kernel Philox keyed "
                                   f"by global row",
                       "log_prob": bool(a.log_prob),
                       "rows_per_gpu": n, "m": m, "sharding": "rows, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel_ms": kern_ms, "kernel_ms_min": float(np.min(kern_all)),
                         "kernel_ms_median": float(np.median(kern_all)), "bytes_per_row": bytes_per_row,
                         "device_copy_GBps": copy_gbps, "frac_of_device_copy": achieved / copy_gbps},
        }
        if world == 1 and a.cpu_rows > 0:
            out["cpu_baseline"] = cpu_baseline(m, a.cpu_rows, seed, a.log_prob)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
