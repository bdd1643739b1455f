#!/usr/bin/env python3
"""Benchmark of the vMF latent-sampling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config C]

N = 1 runs in this process.  N > 1 without a launcher: this process starts N rank processes itself (before anything
touches a GPU), one per GPU, and relays rank 0's JSON line.  Under a launcher (`python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N ...`) RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment.
Every rank does W untimed steps, then K timed steps between barrier + synchronize on both sides; the time is the
maximum over ranks and `value` the work of all ranks divided by it.  Rank 0 prints ONE JSON line.

--config 3 (default, the configuration the headline metric is quoted on): batch vMF rsample + log_prob, m = 64, fp32,
    2^24 rows PER GPU, one (mu_i, kappa_i) per row, mu = normalize(randn), kappa = exp(U[0, ln 1e4]), noise from the
    in-kernel Philox streams keyed by the GLOBAL row index (rank r owns rows [r 2^24, (r+1) 2^24)): weak scaling, no
    data-path collective.  A step is one launch of `vmf_rsample_fwd` (C ABI of libvmf_hip.so) that writes z and,
    fused, log p(z).  `--no-log-prob` times the sampler alone.
--config 4: the same kernel at m = 512, 2^22 rows, kappa up to 1e5.
--config 1: the reference's own CPU-runnable case on the GPU: VonMisesFisher3D(m = 3, kappa in [1, 100]).rsample(4096)
    + mean KL against the uniform prior (two launches per step).
--config 2: Spherical-VAE train step on synthetic 28x28 inputs, batch 128, latent S^9 (forward + ELBO + backward +
    Adam as one HIP graph replay).
--config 5: T^16 = 16 von Mises circles per row, 2^23 rows per GPU (2^27 circle draws), plus the S^1-VAE's
    projected-normal sampler and its I1/I0 KL on the same kappa (three launches per step).

roofline : algorithmic bytes of the dominant kernel per launch / that kernel's average duration, measured with HIP
           events on the launch stream inside the timed region, against 8000 GB/s.
cpu_baseline : the same workload on the host cores of rank 0, bounded sample (N = 1 only): the CPU oracle
           (oracle/libvmf_oracle.so, OpenMP over rows) and, for the m = 3 configurations, the torch-op restatement of
           the reference's own path (oracle/reference_torch.py).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # the defaults are the flags the round-end driver passes.  (After an idle period the GPU needs ~15 launches of the
    # headline size to settle -- per-dispatch times: 1.55, 2.0, 2.1, 1.9 ... then 1.53-1.55 ms -- so the mean of 20 timed
    # steps after 5 warm-up steps sits 2-3 % above the settled time; kernel_ms_median shows the latter.)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, choices=[1, 2, 3, 4, 5])
    ap.add_argument("--rows", type=int, default=None, help="rows per GPU (default: the configuration's own size)")
    ap.add_argument("--m", type=int, default=None, help="ambient dimension for configs 3 / 4")
    ap.add_argument("--log-prob", dest="log_prob", action="store_true", default=True,
                    help="configs 3 / 4: write the fused log p(z) as well (520 B/row at m = 64; the default)")
    ap.add_argument("--no-log-prob", dest="log_prob", action="store_false", help="sampler alone (516 B/row)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank code path on a box with fewer GPUs than ranks)")
    ap.add_argument("--with-allreduce", action="store_true",
                    help="every step also all-reduces the packed fp64 loss scalars (the path's only collective)")
    ap.add_argument("--cpu-rows", type=int, default=None, help="rows of the CPU-baseline sample (0: skip)")
    ap.add_argument("--rehearse", action="store_true",
                    help="launcher / rendezvous / reduction plumbing only: no GPU is touched, the step is empty and the "
                         "JSON line says so (what a CPU-only machine can check)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic_latest.json"),
                    help="file with {'hbm_bytes_per_launch': ...} from a rocprofv3 --pmc pass of the headline kernel")
    return ap.parse_args(argv)


# ---- N > 1 without a launcher: start the ranks ourselves ----------------------------------------------------------------
def launch_ranks(a):
    """Start a.gpus fresh rank processes (this process has made no GPU call and makes none), wait, relay rank 0's
    output; any failing rank fails the run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        # HSA_ENABLE_IPC_MODE_LEGACY: RCCL's intra-node transport shares device buffers between the rank processes
        # through IPC handles.  The hosts this runs on support only the dmabuf flavour of that; with the runtime's legacy
        # mode (its default when the variable is unset) hipIpcGetMemHandle fails with "invalid argument" and the
        # communicator never comes up.  The launching environment normally exports 0 already: an existing value is kept,
        # and only an unset variable is filled in for the ranks started here.
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, rc = "", 0
    deadline = time.time() + 3000
    try:
        out0 = procs[0].communicate(timeout=max(1.0, deadline - time.time()))[0]
        for p in procs[1:]:
            p.wait(timeout=max(1.0, deadline - time.time()))
    except subprocess.TimeoutExpired:
        rc = 124
    for r, p in enumerate(procs):
        if p.poll() is None:
            p.kill()
            rc = rc or 124
        elif p.returncode != 0:
            print(f"bench.py: rank {r} exited with status {p.returncode}", file=sys.stderr)
            rc = rc or (p.returncode if p.returncode > 0 else 1)
    for line in out0.splitlines():  # rank 0's JSON line goes to stdout, whatever else a library printed there to stderr
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr)
    sys.stdout.flush()
    return rc


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def timed_cpu(fn, seconds=10.0, max_passes=200):
    """repeat fn(pass_index) for about `seconds`; returns (passes, elapsed)"""
    t0 = time.perf_counter()
    passes = 0
    while True:
        fn(passes)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or passes >= max_passes:
            return passes, dt


# ---- workloads ---------------------------------------------------------------------------------------------------------
def copy_yardstick(dst, src, min_ms=50.0):
    """what a plain device copy of src into dst reaches on this GPU, GB/s (read + write), the best of the copies made
    in at least `min_ms` of wall time -- before the warm-up steps, outside the timed region.  (After an idle period the
    card runs its first tens of milliseconds of a kernel 5-15 % slower; the yardstick is taken over a stretch long enough
    for the copy itself to settle.  It does not settle the sampler kernels: their first ~25 launches still run slower
    whatever ran before them -- kernel_ms_each in the line shows the transient, kernel_ms_median the settled time.)"""
    best, t0 = None, time.perf_counter()
    dst.copy_(src)
    while best is None or (time.perf_counter() - t0) * 1e3 < min_ms:
        cev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for s_, e_ in cev:
            s_.record()
            dst.copy_(src)
            e_.record()
        torch.cuda.synchronize()
        ms = min(s_.elapsed_time(e_) for s_, e_ in cev)
        best = ms if best is None else min(best, ms)
    return 2 * src.numel() * src.element_size() / (best * 1e-3) / 1e9


class GeneralRSample:
    """configs 3 and 4: vmf_rsample_fwd over the rank's rows"""

    def __init__(self, a, dev, rank, world):
        from medicalimageanalysis_amd import _lib
        self._lib, self.lib = _lib, _lib.lib()
        four = a.config == 4
        self.m = a.m or (512 if four else 64)
        self.n = a.rows or ((1 << 22) if four else (1 << 24))
        self.kmax = 1e5 if four else 1e4
        self.log_prob = a.log_prob
        self.seed, self.row0 = 0x5EED, rank * self.n
        n, m = self.n, self.m
        g = torch.Generator(device=dev).manual_seed(1000 + rank)
        self.mu = torch.empty((n, m), device=dev, dtype=torch.float32)
        chunk = max(1, (1 << 26) // m)
        for s in range(0, n, chunk):  # chunked to keep the temporary small
            e = min(n, s + chunk)
            self.mu[s:e] = torch.nn.functional.normalize(torch.randn(e - s, m, device=dev, generator=g), dim=-1)
        self.kappa = torch.exp(torch.rand(n, device=dev, generator=g) * float(np.log(self.kmax)))
        self.z = torch.empty_like(self.mu)
        self.logp = torch.empty(n, device=dev, dtype=torch.float32) if a.log_prob else None
        self.sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        self.units = n
        self.dtype = "f32"
        self.unit = "samples/s"
        self.bytes_per_unit = (2 * m + 1 + (1 if a.log_prob else 0)) * 4
        self.kernel = "vmf_rsample_fwd_kernel"
        if m == 64 and n == 1 << 24:
            self.metric = "vMF rsample samples/sec (m=64, batch 2^24)"
        else:
            self.metric = f"vMF rsample samples/sec (m={m}, batch {n})"
        self.workload = (f"batch vMF rsample{' + log_prob (fused)' if a.log_prob else ''}, m={m}, {n} rows per GPU, one "
                         f"(mu,kappa) per row, kappa log-uniform [1,{self.kmax:g}], in-kernel Philox keyed by global row")
        self.config = {"log_prob": bool(a.log_prob), "rows_per_gpu": n, "m": m, "sharding": "rows, no data-path collective"}

    def step(self, i, ev):
        _lib = self._lib
        if ev:
            ev[0].record()
        rc = self.lib.vmf_rsample_fwd(_lib.ptr(self.mu), self.m, _lib.ptr(self.kappa), 1, _lib.ptr(self.z), None, None, None,
                                      _lib.ptr(self.logp), None, self.n, self.m, None, None, 0, None, 64, self.row0, self.seed, i,
                                      None, 0, _lib.VMF_F32, self.sp)
        if ev:
            ev[1].record()
        if rc:
            _lib.check(rc, "vmf_rsample_fwd")

    def check(self):
        nrm = self.z[:: max(1, self.n // 4096)].norm(dim=-1)
        assert torch.isfinite(nrm).all() and float((nrm - 1).abs().max()) < 1e-3, "rsample produced non-unit rows"

    def copy_GBps(self):
        """what a plain device copy of the same 2 x n x m x 4 bytes reaches on this GPU (before the warm-up steps, outside
        the timed region)"""
        return copy_yardstick(self.z, self.mu)

    def cpu_baseline(self, rows):
        from oracle import vmf_oracle as orc
        rows = rows if rows is not None else max(4096, (1 << 28) // self.m)
        gen = np.random.default_rng(1234)
        mu = gen.standard_normal((rows, self.m), dtype=np.float32)
        mu /= np.linalg.norm(mu, axis=1, keepdims=True)
        kappa = np.exp(gen.uniform(0, np.log(self.kmax), rows)).astype(np.float32)
        orc.vmf_rsample_philox(mu[:4096], kappa[:4096], seed=self.seed)  # warm-up (loads the library, spins up threads)

        def one(p):  # a new noise offset each pass
            z, _, _ = orc.vmf_rsample_philox(mu, kappa, seed=self.seed, offset=p, row0=0, max_prop=64)
            if self.log_prob:
                orc.vmf_log_prob(mu, kappa, z)
        passes, dt = timed_cpu(one)
        return {"value": rows * passes / dt, "unit": "samples/s", "cores": int(orc.lib().orc_num_threads()),
                "kind": "port", "cpu_model": cpu_model(), "host_cpus": os.cpu_count(),
                "sample": f"{passes} passes over {rows} rows of the same workload (m={self.m}, fp32, oracle's Philox"
                          + (", + log_prob" if self.log_prob else "") + f"), {dt:.1f} s"}


class Reference3D:
    """config 1: VonMisesFisher3D.rsample(B) + mean KL(vMF || uniform), the two launches the reference's
    sampler/von_mises_fisher.py:66-86 and model/spherical_vae.py:141-184 (+ .mean()) become"""

    def __init__(self, a, dev, rank, world):
        from medicalimageanalysis_amd import _lib
        from medicalimageanalysis_amd.sampler.von_mises_fisher import vmf3_broadcast_plan
        self._lib, self.lib = _lib, _lib.lib()
        self.n = n = a.rows or 4096
        g = torch.Generator(device=dev).manual_seed(1000 + rank)
        self.loc = torch.randn(n, 3, device=dev, generator=g)
        self.kappa = torch.rand(n, device=dev, generator=g) * 99 + 1
        self.z = torch.empty_like(self.loc)
        self.kl = torch.empty(1, device=dev, dtype=torch.float64)
        self.ws = torch.empty(_lib.KL_WORKSPACE_DOUBLES, device=dev, dtype=torch.float64)
        plan = vmf3_broadcast_plan((n, 3), (n,))
        self.nd = len(plan["odims"])
        self.pa = [_lib.i64_array(plan[k]) for k in ("odims", "w_str", "k_str", "l_str", "e_str")]
        self.sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        self.seed, self.row0 = 0x5EED, rank * n  # one batch sharded by rows: the in-kernel noise is keyed by the global row
        self.units, self.dtype, self.unit = n, "f32", "samples/s"
        self.bytes_per_unit = 28
        self.kernel = "vmf3_rows_kernel"
        self.metric = f"VonMisesFisher3D rsample + KL samples/sec (m=3, batch {n})"
        self.workload = (f"VonMisesFisher3D(loc, kappa in [1,100]).rsample() of {n} rows per GPU (in-kernel Philox) + mean "
                         f"KL against the uniform prior (spherical_vae's piecewise form), two launches per step")
        self.config = {"rows_per_gpu": n, "m": 3, "sharding": "rows, no data-path collective"}

    def step(self, i, ev):
        _lib = self._lib
        if ev:
            ev[0].record()
        rc = self.lib.vmf3_rsample_fwd(_lib.ptr(self.loc), _lib.ptr(self.kappa), None, None, _lib.ptr(self.z), None, None,
                                       self.nd, *self.pa, 0, _lib.VMF_F32, self.row0, self.seed, i, None, self.sp)
        if ev:
            ev[1].record()
        rc = rc or self.lib.vmf_kl_fwd(_lib.ptr(self.kappa), None, _lib.ptr(self.kl), _lib.ptr(self.ws), 1.0 / self.n, self.n,
                                       _lib.KL_SPHERICAL_PIECEWISE, _lib.VMF_F32, self.sp)
        if rc:
            _lib.check(rc, "vmf3_rsample_fwd / vmf_kl_fwd")

    def check(self):
        nrm = self.z.norm(dim=-1)
        assert float((nrm - 1).abs().max()) < 1e-3 and 0.0 < float(self.kl) < 10.0

    def copy_GBps(self):
        return None

    def cpu_baseline(self, rows):
        """the reference's own op sequence in torch on the host cores (restated in oracle/reference_torch.py), and the
        oracle's C port beside it"""
        from oracle import reference_torch as rt
        from oracle import vmf_oracle as orc
        n = rows if rows is not None else self.n
        torch.manual_seed(0)
        loc = torch.randn(n, 3)
        kappa = torch.rand(n) * 99 + 1
        rt.rsample(loc, kappa)

        def one(p):
            rt.rsample(loc, kappa)
            rt.kl_spherical(kappa).mean()
        passes, dt = timed_cpu(one, seconds=5.0, max_passes=20000)
        locn, kn = loc.numpy(), kappa.numpy()
        gen = np.random.default_rng(0)
        u, eps = gen.uniform(size=n).astype(np.float32), gen.standard_normal((n, 3)).astype(np.float32)

        def port(p):
            orc.vmf3_rsample_fwd(locn, kn, u, eps)
            orc.vmf_kl(kn, 0)
        p2, dt2 = timed_cpu(port, seconds=3.0, max_passes=20000)
        return {"value": n * passes / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                "cpu_model": cpu_model(), "host_cpus": os.cpu_count(),
                "sample": f"{passes} passes of rsample({n}) + KL.mean() in torch ops on the host, fp32, {dt:.1f} s "
                          f"(the reference's op sequence: sampler/von_mises_fisher.py:50-86, model/spherical_vae.py:141-184)",
                "oracle_port_value": n * p2 / dt2, "oracle_port_note": "C oracle with given noise, no RNG in the loop"}


class VAETrainStep:
    """config 2: Spherical-VAE train step, 28x28 inputs, batch 128, latent S^9 (general-m sampler, m = 10)"""

    def __init__(self, a, dev, rank, world):
        from medicalimageanalysis_amd.graph import GraphedTrainStep
        from medicalimageanalysis_amd.model.hyperspherical_vae import HypersphericalVAE
        self.B = B = a.rows or 128
        self.cfg = dict(extrinsic_dim=784, latent_dim=10, sftbeta=1.0, encoder_width=128, encoder_depth=4,
                        decoder_width=128, decoder_depth=4, dropout_p=0.1, device=str(dev), gamma=1.0, beta=0.01)
        torch.manual_seed(rank)
        self.model = HypersphericalVAE(self.cfg)
        self.opt = torch.optim.Adam(self.model.parameters(), lr=1e-3, capturable=True, fused=True)
        self.x = torch.rand(B, 784, device=dev)
        torch.distributions.Distribution.set_default_validate_args(False)

        def loss_fn(mdl, xb):
            z, x_mu, params = mdl(xb)
            return mdl._elbo(xb, x_mu, params)
        self.loss_fn = loss_fn
        self.graphed = GraphedTrainStep(self.model, self.opt, loss_fn, [self.x], warmup=3, tune_gemms=True)
        self.loss = None
        self.units, self.dtype, self.unit = B, "f32", "samples/s"
        self.bytes_per_unit = (2 * 10 + 1) * 4
        self.kernel = "vmf_rsample_fwd_kernel (m=10)"
        self.metric = f"Spherical VAE train step samples/sec (28x28, batch {B}, latent S^9)"
        self.workload = (f"HypersphericalVAE (784-128x5-S^9-128x5-784, softplus MLPs) forward + ELBO + backward + Adam on a "
                         f"synthetic batch of {B}, one HIP graph replay per step (fused Adam update; the nn.Linear GEMM kernels chosen "
                         f"by PyTorch's TunableOp during warm-up), fp32, independent replica per GPU")
        self.config = {"batch_per_gpu": B, "latent_dim": 10, "sharding": "replicas (no gradient exchange)"}

    def step(self, i, ev):
        self.loss = self.graphed(self.x)

    def check(self):
        assert torch.isfinite(self.loss).all()

    def copy_GBps(self):
        return None

    def sampler_kernel_ms(self, dev):
        """the path's own kernel at this shape, timed by itself (a graph replay cannot be bracketed from the host)"""
        from medicalimageanalysis_amd import _lib
        mu = torch.nn.functional.normalize(torch.randn(self.B, 10, device=dev), dim=-1)
        kap = torch.rand(self.B, device=dev) * 50 + 1
        z = torch.empty_like(mu)
        sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for i in range(30):
            if i >= 10:
                ev[i - 10][0].record()
            _lib.lib().vmf_rsample_fwd(_lib.ptr(mu), 10, _lib.ptr(kap), 1, _lib.ptr(z), None, None, None, None, None, self.B, 10,
                                       None, None, 0, None, 64, 0, 1, i, None, 0, _lib.VMF_F32, sp)
            if i >= 10:
                ev[i - 10][1].record()
        torch.cuda.synchronize()
        return float(np.mean([s.elapsed_time(e) for s, e in ev]))

    def eager_ms(self):
        def one():
            self.opt.zero_grad(set_to_none=True)
            loss = self.loss_fn(self.model, self.x)
            loss.backward()
            self.opt.step()
        for _ in range(5):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            one()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 20 * 1e3

    def cpu_baseline(self, rows):
        from oracle import reference_torch as rt
        cfg = dict(self.cfg, latent_dim=3, device="cpu")
        torch.manual_seed(0)
        model = rt.SphericalVAERef(cfg)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        x = torch.rand(self.B, 784)

        def one(p):
            z, x_mu, params = model(x)
            loss = model.elbo(x, x_mu, params)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
        one(0)
        passes, dt = timed_cpu(one, seconds=10.0, max_passes=5000)
        return {"value": self.B * passes / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                "cpu_model": cpu_model(), "host_cpus": os.cpu_count(),
                "sample": f"{passes} train steps of the reference-equivalent Spherical VAE in torch ops on the host, batch "
                          f"{self.B}, latent_dim 3 (the reference's sampler asserts m == 3, so z is ({self.B},{self.B},3) "
                          f"and the decoder runs {self.B * self.B} rows), {dt:.1f} s"}


class TorusAndS1:
    """config 5: T^16 (16 von Mises circles per row) + the S^1-VAE's projected-normal sampler and I1/I0 KL"""

    def __init__(self, a, dev, rank, world):
        from medicalimageanalysis_amd import _lib
        self._lib, self.lib = _lib, _lib.lib()
        self.rows = a.rows or (1 << 23)
        self.n = n = self.rows * 16
        g = torch.Generator(device=dev).manual_seed(1000 + rank)
        ang = torch.rand(n, device=dev, generator=g) * (2 * np.pi)
        self.mu = torch.stack([ang.cos(), ang.sin()], dim=-1).contiguous()
        del ang
        self.kappa = torch.exp(torch.rand(n, device=dev, generator=g) * float(np.log(1e3)))
        self.z = torch.empty_like(self.mu)
        self.z2 = torch.empty_like(self.mu)
        self.kl = torch.empty(1, device=dev, dtype=torch.float64)
        self.ws = torch.empty(_lib.KL_WORKSPACE_DOUBLES, device=dev, dtype=torch.float64)
        self.sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        self.seed, self.row0 = 0x5EED, rank * n
        self.units, self.dtype, self.unit = n, "f32", "circle draws/s"
        self.bytes_per_unit = 20
        self.kernel = "vmf_circle_fwd_kernel"
        self.metric = f"T^16 von Mises draws/sec ({self.rows} rows x 16 circles per GPU) + S^1-VAE sampler and KL"
        self.workload = (f"product of 16 von Mises circles: vMF m=2 rsample over {n} (mu,kappa) pairs per GPU, kappa "
                         f"log-uniform [1,1e3], then the S^1-VAE projected-normal sampler and the sum of kappa I1/I0 - log I0 "
                         f"over the same kappa; three launches per step, in-kernel Philox keyed by global row")
        self.config = {"rows_per_gpu": self.rows, "circles": 16, "m": 2, "sharding": "rows, no data-path collective"}

    def step(self, i, ev):
        _lib, n = self._lib, self.n
        if ev:
            ev[0].record()
        rc = self.lib.vmf_rsample_fwd(_lib.ptr(self.mu), 2, _lib.ptr(self.kappa), 1, _lib.ptr(self.z), None, None, None, None,
                                      None, n, 2, None, None, 0, None, 64, self.row0, self.seed, 2 * i, None, 0, _lib.VMF_F32,
                                      self.sp)
        if ev:
            ev[1].record()
        rc = rc or self.lib.vmf_s1_projnormal_fwd(_lib.ptr(self.mu), _lib.ptr(self.kappa), None, _lib.ptr(self.z2), n, 2,
                                                  self.row0, self.seed, 2 * i + 1, None, _lib.VMF_F32, self.sp)
        rc = rc or self.lib.vmf_kl_fwd(_lib.ptr(self.kappa), None, _lib.ptr(self.kl), _lib.ptr(self.ws), 1.0, n,
                                       _lib.KL_STABLE_M2, _lib.VMF_F32, self.sp)
        if rc:
            _lib.check(rc, "config 5 step")

    def check(self):
        s = slice(None, None, max(1, self.n // 4096))
        for t in (self.z, self.z2):
            assert float((t[s].norm(dim=-1) - 1).abs().max()) < 1e-3
        assert np.isfinite(float(self.kl))

    def copy_GBps(self):
        """a plain device copy of the circle means into the sample buffer (8 B read + 8 B written per pair)"""
        return copy_yardstick(self.z, self.mu)

    def cpu_baseline(self, rows):
        from oracle import vmf_oracle as orc
        n = rows if rows is not None else 1 << 22
        gen = np.random.default_rng(1234)
        ang = gen.uniform(0, 2 * np.pi, n)
        mu = np.stack([np.cos(ang), np.sin(ang)], axis=-1).astype(np.float32)
        kappa = np.exp(gen.uniform(0, np.log(1e3), n)).astype(np.float32)
        noise = gen.standard_normal((n, 2)).astype(np.float32)

        def one(p):
            orc.vmf_rsample_philox(mu, kappa, seed=self.seed, offset=p, row0=0, max_prop=64)
            orc.s1_projnormal_fwd(mu, kappa, noise)
            orc.vmf_kl(kappa, 5)
        one(0)
        passes, dt = timed_cpu(one)
        return {"value": n * passes / dt, "unit": "circle draws/s", "cores": int(orc.lib().orc_num_threads()),
                "kind": "port", "cpu_model": cpu_model(), "host_cpus": os.cpu_count(),
                "sample": f"{passes} passes over {n} (mu,kappa) pairs of the same three stages (oracle; the projected-normal "
                          f"stage with pre-drawn normals), {dt:.1f} s"}


class Rehearsal:
    """--rehearse: no device work.  Exercises the rank launcher, the rendezvous, the barriers and the reductions of the
    timing code on any machine."""

    def __init__(self, a, dev, rank, world):
        self.units, self.dtype, self.unit = a.rows or 1, "none", "units/s"
        self.bytes_per_unit, self.kernel = 0, "none"
        self.metric, self.workload = "rehearsal (no GPU work)", "empty step"
        self.config = {"rows_per_gpu": self.units}

    def step(self, i, ev):
        pass

    def check(self):
        pass

    def copy_GBps(self):
        return None


WORKLOADS = {1: Reference3D, 2: VAETrainStep, 3: GeneralRSample, 4: GeneralRSample, 5: TorusAndS1}


def run_rank(a):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # VMF_BENCH_FORCE_DIST=1 runs the process-group path (RCCL init, barriers, the max-reduce of the time) with a single
    # rank too: the rehearsal of the multi-GPU code path that fits a one-GPU box
    use_dist = world > 1 or os.environ.get("VMF_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
    gpu = not a.rehearse
    if gpu:
        local = local % max(1, torch.cuda.device_count())  # more ranks than GPUs (rehearsal on a small box): share
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    else:
        dev = torch.device("cpu")
    backend = "gloo" if a.rehearse else a.backend
    if use_dist:
        # RCCL and gloo print banners ("RCCL version : ...", "[Gloo] Rank 0 is connected ...") on the process's stdout when
        # the communicator comes up, which happens at the first collective: keep stdout for the JSON line
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            dist.all_reduce(torch.zeros(1, device=dev if backend == "nccl" else "cpu"))
            if gpu:
                torch.cuda.synchronize(dev)
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)
    ranks_seen = dist.get_world_size() if use_dist else 1
    # which GPU this rank really runs on (ordinal + PCI address + uuid): an N-rank line must show N distinct devices
    if gpu:
        pr = torch.cuda.get_device_properties(dev)
        pci = "{:04x}:{:02x}:{:02x}".format(*(int(getattr(pr, k, -1)) & 0xffff for k in
                                             ("pci_domain_id", "pci_bus_id", "pci_device_id")))
        my_dev = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "ordinal": local, "pci": pci,
                  "uuid": str(getattr(pr, "uuid", "")), "name": pr.name,
                  "visible": os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES"))}
    else:
        my_dev = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "ordinal": None, "pci": None,
                  "pid": os.getpid()}
    if a.gpus != world and rank == 0:
        print(f"warning: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    cdev = dev if backend == "nccl" else torch.device("cpu")  # where collectives on scalars live

    def sync():
        if gpu:
            torch.cuda.synchronize(dev)

    wl = (Rehearsal if a.rehearse else WORKLOADS[a.config])(a, dev, rank, world)
    # the path's only collective: one packed fp64 buffer [sum KL, sum recon, sum log p, count] per step
    scalars = torch.zeros(4, device=cdev, dtype=torch.float64)
    ar_ev = []

    def allreduce(timed):
        scalars[3] = float(wl.units)
        if gpu and timed and backend == "nccl":
            e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            e[0].record()
            dist.all_reduce(scalars)
            e[1].record()
            ar_ev.append(e)
        else:
            dist.all_reduce(scalars)

    do_ar = a.with_allreduce and use_dist
    # The roofline's yardstick -- a plain device copy of the same buffers -- is measured first, while the output buffer is
    # still scratch.  (tools/clock_probe.py: from idle, launches 2..12 of the headline kernel run 1.6-2.1 ms before
    # settling; with the copies in front they run 1.54-1.55, and reach the settled 1.48-1.49 after ~25 launches.)
    copy = wl.copy_GBps() if gpu else None
    for i in range(a.warmup):
        wl.step(i, None)
        if do_ar:
            allreduce(False)
    sync()
    if use_dist:
        dist.barrier()
    sync()
    evs = ([(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
           if gpu else [None] * a.steps)
    t0 = time.perf_counter()
    for i in range(a.steps):
        wl.step(a.warmup + i, evs[i])
        if do_ar:
            allreduce(True)
    sync()
    if use_dist:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    kern_all = [s.elapsed_time(e) for s, e in evs] if gpu and a.config != 2 else []
    kern_ms = float(np.mean(kern_all)) if kern_all else None
    per_rank_kernel_ms = [kern_ms]
    if use_dist:
        t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([kern_ms if kern_ms is not None else -1.0], device=cdev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(ranks_seen)]
        dist.all_gather(every, mine)
        per_rank_kernel_ms = [float(x.item()) if float(x.item()) >= 0 else None for x in every]
        devices = [None] * ranks_seen
        dist.all_gather_object(devices, my_dev)
    else:
        devices = [my_dev]
    wl.check()

    if rank == 0:
        out = {
            "metric": wl.metric,
            "value": world * wl.units * a.steps / elapsed if not a.rehearse else 0.0,
            "unit": wl.unit,
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": wl.dtype,
            "data": "synthetic",
            "config": dict({"workload": wl.workload, "bench_config": a.config}, **wl.config),
            "ranks_seen": ranks_seen,
            "kernel_ms_per_rank": per_rank_kernel_ms,
            "devices": devices,
            "distinct_devices": len({(d.get("pci"), d.get("uuid")) for d in devices}) if gpu else None,
        }
        if a.rehearse:
            out["rehearsal"] = True
        if do_ar:
            out["allreduce"] = {"what": "packed fp64 [sum KL, sum recon, sum log p, count] all-reduce after every step",
                                "backend": backend, "count_seen": float(scalars[3].item()),
                                "ms": float(np.mean([s.elapsed_time(e) for s, e in ar_ev])) if ar_ev else None}
        if gpu:
            if a.config == 2:  # a graph replay cannot be bracketed: the sampler kernel is timed by itself afterwards
                kern_ms = wl.sampler_kernel_ms(dev)
                kern_all = [kern_ms]
                out["eager_ms_per_step"] = wl.eager_ms()
            nbytes = wl.units * wl.bytes_per_unit
            achieved = nbytes / (kern_ms * 1e-3) / 1e9
            traffic, traffic_source = None, None
            if a.config == 3:
                try:
                    with open(a.traffic_json) as f:
                        tj = json.load(f)
                    if (tj.get("rows") == wl.n and tj.get("m") == wl.m
                            and bool(tj.get("log_prob", False)) == bool(a.log_prob)):
                        traffic = tj.get("hbm_bytes_per_launch")
                        traffic_source = ("rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE) of this kernel on this "
                                          "workload, recorded in " + os.path.relpath(a.traffic_json, ROOT)
                                          + "; not re-measured in this run")
                except (OSError, ValueError):
                    pass
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                               "kernel": wl.kernel, "kernel_ms": kern_ms, "kernel_ms_min": float(np.min(kern_all)),
                               "kernel_ms_median": float(np.median(kern_all)),
                               "kernel_ms_each": [round(float(t), 4) for t in kern_all],
                               "achieved_at_median": nbytes / (float(np.median(kern_all)) * 1e-3) / 1e9,
                               "bytes_per_unit": wl.bytes_per_unit, "bytes_per_launch": nbytes,
                               "device_copy_GBps": copy, "frac_of_device_copy": achieved / copy if copy else None}
            if a.config == 2:
                out["roofline"]["measured"] = "sampler kernel alone at the step's shape, outside the timed region"
            if world == 1 and a.cpu_rows != 0:
                out["cpu_baseline"] = wl.cpu_baseline(a.cpu_rows)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a)
    return run_rank(a)


if __name__ == "__main__":
    sys.exit(main())
