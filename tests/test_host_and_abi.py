"""CPU-side checks: the C ABI library loads and exports every symbol include/vmf_hip.h declares, host logic
(broadcast plans, error conventions, generator state, module aliases) and the multi-rank path on gloo."""
import ctypes as C
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    from medicalimageanalysis_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "vmf_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(\w+)\(", hdr, flags=re.M))
    assert len(declared) >= 20
    assert declared == set(_lib.EXPORTS)
    h = C.CDLL(_lib.library_path())
    for name in declared:
        assert hasattr(h, name), name
    lib = _lib.lib()
    assert lib.vmf_version() == 141
    # argument errors are reported without touching a GPU
    assert lib.vmf_rsample_fwd(None, 64, None, 1, None, None, None, None, None, None, -1, 64, None, None, 0, None, 64, 0, 0, 0, None, 0,
                               0, None) == -1
    assert b"n < 0" in lib.vmf_last_error_string()
    assert lib.vmf_rsample_fwd(None, 64, None, 1, None, None, None, None, None, None, 5, 4096, None, None, 0, None, 64, 0, 0, 0, None, 0,
                               0, None) == -4
    assert lib.vmf_kl_fwd(None, None, None, None, 1.0, 0, 99, 0, None) == -1


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "medicalimageanalysis_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle" not in src.lower().replace("no cpu", ""), os.path.join(dp, f)


@pytest.mark.parametrize("loc,scale,ss", [((33, 3), (33,), ()), ((33, 3), (33, 1), ()), ((19, 3), (19,), (5,)),
                                          ((4, 3), (4, 1), (4,)), ((2, 3, 3), (2, 3), ()), ((3,), (), ()),
                                          ((7, 3), (1,), (2, 3)), ((1, 3), (6, 1), ())])
def test_broadcast_plan_matches_numpy_broadcasting(oracle, loc, scale, ss):
    from medicalimageanalysis_amd.sampler.von_mises_fisher import vmf3_broadcast_plan

    mine = vmf3_broadcast_plan(loc, scale, ss)
    ref = oracle.vmf3_plan(loc, scale, ss)
    assert mine["o_shape"] == ref["o_shape"] and mine["w_shape"] == ref["w_shape"] and mine["e_shape"] == ref["e_shape"]
    odims = mine["odims"]
    n = int(np.prod(odims))
    # same operand index for every output row (strides may differ on size-1 dims)
    idx = np.stack(np.unravel_index(np.arange(n), odims), axis=-1) if n else np.zeros((0, len(odims)), dtype=int)
    for key in ("w_str", "k_str", "l_str", "e_str"):
        a = idx @ np.array(mine[key]).reshape(-1)
        r = np.array(ref[key]) if len(ref[key]) else np.zeros(1, dtype=int)
        b = idx @ r.reshape(-1)
        assert np.array_equal(a, b), key


@pytest.mark.parametrize("loc,scale,ss,kind,ni,nj", [
    ((33, 3), (33,), (), 1, 33, 33),          # CircularVAE: kappa squeezed to (B,)
    ((33, 3), (33, 1), (), 2, 33, 33),        # SphericalVAE / ToroidalVAE: (B, 1) kappa -> (B, B, 3)
    ((2, 3, 3), (2, 3), (), 1, 6, 6),         # batch dims collapse to rows
    ((5, 3), (7, 1), (), 2, 7, 5),            # outer product with different sizes
    ((19, 3), (19,), (5,), 0, 0, 0),          # sample shape in front of shared parameters
    ((4, 3), (4, 1), (4,), 0, 0, 0),
    ((3,), (), (), 1, 1, 1),
    ((1, 3), (6, 1), (), 0, 0, 0),            # one loc row against six scales: eps has one row too, no outer product
])
def test_broadcast_kind_classification(loc, scale, ss, kind, ni, nj):
    """which kernel a call takes is decided on the host from the broadcast description alone"""
    from medicalimageanalysis_amd import _lib
    from medicalimageanalysis_amd.sampler.von_mises_fisher import _plan_kind, vmf3_broadcast_plan

    got = _plan_kind(vmf3_broadcast_plan(loc, scale, ss))
    assert got[0] == kind
    if kind:
        assert got[1:] == (ni, nj)
    assert (_lib.KIND_GENERAL, _lib.KIND_ROWS, _lib.KIND_OUTER) == (0, 1, 2)


def test_outer_backward_workspace_size_is_reported_without_a_gpu():
    from medicalimageanalysis_amd import _lib

    lib = _lib.lib()
    nb = C.c_int64(-1)
    assert lib.vmf3_rsample_bwd_outer_workspace(4096, 4096, _lib.VMF_F32, C.byref(nb)) == 0
    # one (a, b) pair per row and 256-column tile, one (dl, dv) pair per column and 64-row tile: 3.4 % of grad z
    assert nb.value == (16 * 4096 * 2 + 64 * 4096 * 6) * 4
    assert lib.vmf3_rsample_bwd_outer_workspace(128, 128, _lib.VMF_F64, C.byref(nb)) == 0 and nb.value > 0
    assert lib.vmf3_rsample_bwd_outer_workspace(-1, 4, _lib.VMF_F32, C.byref(nb)) == -1
    # too small a workspace is refused before anything is launched
    rc = lib.vmf3_rsample_bwd_outer(C.c_void_p(16), C.c_void_p(16), None, None, C.c_void_p(16), None, C.c_void_p(16),
                                    C.c_void_p(16), 64, 64, C.c_void_p(16), 8, 0, 0, 0, 0, 0, None, None)
    assert rc == -1 and b"workspace" in lib.vmf_last_error_string()


def test_broadcast_plan_rejects_incompatible_shapes():
    from medicalimageanalysis_amd.sampler.von_mises_fisher import vmf3_broadcast_plan

    with pytest.raises(RuntimeError):
        vmf3_broadcast_plan((5, 3), (4, 1), (3,))  # (3,4,1) vs (3,5): the reference's `*` raises too


def test_error_conventions_without_gpu():
    from medicalimageanalysis_amd.sampler.spherical_uniform import SphericalUniform
    from medicalimageanalysis_amd.sampler.von_mises_fisher import VonMisesFisher, VonMisesFisher3D
    from medicalimageanalysis_amd.sampler.ive import ive

    with pytest.raises(AssertionError, match="This simple version o"):
        VonMisesFisher3D(torch.randn(4, 10), torch.ones(4))
    with pytest.raises(AssertionError, match="at least 2"):
        SphericalUniform(1)
    with pytest.raises(ValueError):
        VonMisesFisher3D(torch.randn(4, 3), -torch.ones(4), validate_args=True)
    q = VonMisesFisher3D(torch.randn(4, 3), torch.ones(4))
    assert q.has_rsample and q.batch_shape == (4,) and callable(q.mean) and callable(q.stddev)
    np.testing.assert_allclose(q.loc.norm(dim=-1).numpy(), 1.0, atol=2e-4)
    with pytest.raises(NotImplementedError):
        q.log_prob(torch.randn(4, 3))
    with pytest.raises(NotImplementedError):
        q.entropy()
    # no CPU path: a CPU tensor is an error, never a silent fallback
    with pytest.raises(RuntimeError, match="MI355X only"):
        q.rsample()
    with pytest.raises(RuntimeError, match="MI355X only"):
        ive(0.5, torch.ones(3))
    with pytest.raises(RuntimeError, match="MI355X only"):
        VonMisesFisher(torch.randn(4, 8), torch.ones(4)).rsample()
    s = SphericalUniform(3)
    assert s.event_shape == (3,) and not s.has_rsample


def test_missing_library_fails_loudly(tmp_path):
    code = ("import os; os.environ['VMF_HIP_LIB']=%r\n"
            "from medicalimageanalysis_amd import _lib\n"
            "try:\n    _lib.lib()\nexcept RuntimeError as e:\n    print('RAISED', 'no CPU fallback' in str(e))\n"
            ) % str(tmp_path / "nope.so")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert "RAISED True" in out.stdout, out.stdout + out.stderr


def test_generator_state_roundtrip():
    from medicalimageanalysis_amd.rng import PhiloxGenerator

    g = PhiloxGenerator(seed=7)
    assert g.next_offset() == (7, 0) and g.next_offset() == (7, 1)
    st = g.get_state()
    g.next_offset()
    g.set_state(st)
    assert g.next_offset() == (7, 2)
    g.manual_seed(9)
    assert g.next_offset() == (9, 0)


def _alias_sandbox():
    """snapshot of the module names install_aliases() may touch"""
    names = ("sampler", "model", "dataloader", "von_mises_fisher", "ive", "circular_VAE", "s1_vae", "geometry")
    return {k: v for k, v in sys.modules.items() if k.split(".")[0] in names}, names


def _alias_restore(saved, names):
    for k in [k for k in sys.modules if k.split(".")[0] in names]:
        del sys.modules[k]
    sys.modules.update(saved)


def test_reference_module_names_resolve():
    import medicalimageanalysis_amd as pkg

    saved, names = _alias_sandbox()
    try:
        pkg.install_aliases()
        ns = {}
        exec("from sampler.von_mises_fisher import *\nfrom sampler.spherical_uniform import *\n"
             "from sampler.ive import ive, Ive, IveFunction\nfrom model.spherical_vae import SphericalVAE\n"
             "from model.torus_vae import ToroidalVAE\n", ns)
        # the reference's model files rely on these names arriving through the star import
        for name in ("math", "torch", "ive", "Distribution", "constraints", "VonMisesFisher3D", "SphericalUniform"):
            assert name in ns, name
        assert "von_mises_fisher" not in sys.modules and "ive" not in sys.modules  # generic names are opt-in
    finally:
        _alias_restore(saved, names)


def test_aliases_leave_the_callers_own_model_package_alone(tmp_path):
    """main.py:1-6 and model/train.py:1-2, model/infer.py:1-3 of the reference import `model.train`, `model.infer`
    and `dataloader.utils` next to the two model files this package replaces: after install_aliases() the caller's
    own `model/` directory must still serve the files that are not replaced."""
    import medicalimageanalysis_amd as pkg

    (tmp_path / "model").mkdir()
    (tmp_path / "dataloader").mkdir()
    (tmp_path / "dataloader" / "data.py").write_text("DATA_MARK = 1\n")
    (tmp_path / "dataloader" / "utils.py").write_text("from dataloader.data import *\nUTILS_MARK = 2\n")
    # the import lines of the reference's model/train.py:1-2 and model/infer.py:1-3, then one own name each
    (tmp_path / "model" / "train.py").write_text(
        "from model.spherical_vae import *\nfrom dataloader.utils import * \nTRAIN_MARK = SphericalVAE.__module__\n")
    (tmp_path / "model" / "infer.py").write_text(
        "from model.spherical_vae import *\nfrom dataloader.utils import * \nfrom model.torus_vae import *\n"
        "INFER_MARK = ToroidalVAE.__module__\n")
    # a stale copy of a replaced file in the caller's tree must lose against the alias
    (tmp_path / "model" / "spherical_vae.py").write_text("raise ImportError('the reference file was imported')\n")
    saved, names = _alias_sandbox()
    _alias_restore({}, names)
    sys.path.insert(0, str(tmp_path))
    try:
        pkg.install_aliases()
        ns = {}
        exec("from dataloader.data import *\nfrom dataloader.utils import *\nfrom model.torus_vae import *\n"
             "from model.train import *\nfrom model.spherical_vae import *\nfrom model.infer import *\n", ns)  # main.py:1-6
        assert ns["TRAIN_MARK"] == "medicalimageanalysis_amd.model.spherical_vae"
        assert ns["INFER_MARK"] == "medicalimageanalysis_amd.model.torus_vae"
        assert ns["DATA_MARK"] == 1 and ns["UTILS_MARK"] == 2
        assert sys.modules["model.train"].__file__.startswith(str(tmp_path))
        import model
        assert model.spherical_vae is sys.modules["medicalimageanalysis_amd.model.spherical_vae"]
    finally:
        sys.path.remove(str(tmp_path))
        _alias_restore(saved, names)


def test_aliases_do_not_hide_a_broken_package_of_the_caller(tmp_path):
    """a caller whose own `model/__init__.py` imports something that is not installed must see that error: the alias
    installer only creates an empty `model` package when no such package exists at all; and a package the caller has
    already imported is used as it is, not imported a second time"""
    import medicalimageanalysis_amd as pkg

    (tmp_path / "model").mkdir()
    (tmp_path / "model" / "__init__.py").write_text("import a_dependency_that_is_not_installed" + chr(10))
    (tmp_path / "sampler").mkdir()
    (tmp_path / "sampler" / "__init__.py").write_text(
        "import builtins; builtins._vmf_test_sampler_inits = getattr(builtins, '_vmf_test_sampler_inits', 0) + 1" + chr(10))
    saved, names = _alias_sandbox()
    _alias_restore({}, names)
    sys.path.insert(0, str(tmp_path))
    import builtins
    try:
        import sampler  # noqa: F401  the caller's own package, imported before the aliases go in
        assert builtins._vmf_test_sampler_inits == 1
        with pytest.raises(ModuleNotFoundError, match="a_dependency_that_is_not_installed"):
            pkg.install_aliases()
        assert builtins._vmf_test_sampler_inits == 1   # not executed again
        assert sys.modules["sampler"].__file__.startswith(str(tmp_path))
    finally:
        sys.path.remove(str(tmp_path))
        if hasattr(builtins, "_vmf_test_sampler_inits"):
            del builtins._vmf_test_sampler_inits
        _alias_restore(saved, names)


def test_flat_module_names_of_the_cnn_and_s1_experiments():
    """CNN_experiment/circular_VAE.py:5 (`from von_mises_fisher import *`), CNN_experiment/von_mises_fisher.py:3
    (`from ive import ive`), CNN_experiment/train_experiment.py:9, regular_cnn/main.py:11-12,
    invariant_cnn/main_invariant.py:10-11"""
    import medicalimageanalysis_amd as pkg
    from medicalimageanalysis_amd.sampler.von_mises_fisher import VonMisesFisher3DLogProb

    saved, names = _alias_sandbox()
    try:
        pkg.install_aliases(flat=True)
        ns = {}
        exec("from von_mises_fisher import *\nfrom ive import ive\nfrom circular_VAE import CircularVAE\n"
             "from s1_vae import S1_VAE, loss_function\nfrom geometry import compute_curvature_profile\n", ns)
        assert ns["compute_curvature_profile"].__module__ == "medicalimageanalysis_amd.geometry"
        assert ns["VonMisesFisher3D"] is VonMisesFisher3DLogProb and hasattr(ns["VonMisesFisher3D"], "log_prob")
        for name in ("math", "torch", "ive", "Distribution", "constraints", "CircularVAE", "S1_VAE", "loss_function"):
            assert name in ns, name
        import ive as flat_ive
        assert {"IveFunction", "Ive", "ive"} <= set(dir(flat_ive))
    finally:
        _alias_restore(saved, names)


def test_shard_rows_cover_and_order():
    from medicalimageanalysis_amd.distributed import shard_rows

    for n in (0, 1, 7, 1 << 24, (1 << 24) + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_rows(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from medicalimageanalysis_amd.distributed import shard_rows, allreduce_scalars, global_mean
from oracle import vmf_oracle as orc
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n, m = 1001, 16
gen = np.random.default_rng(0)
mu = gen.standard_normal((n, m)); mu /= np.linalg.norm(mu, axis=1, keepdims=True)
kappa = np.exp(gen.uniform(0, np.log(1e3), n))
a, b = shard_rows(n, rank, world)
# each rank samples only its rows, numbered globally
z, w, idx = orc.vmf_rsample_philox(mu[a:b], kappa[a:b], seed=77, offset=2, row0=a)
zf, wf, _ = orc.vmf_rsample_philox(mu, kappa, seed=77, offset=2, row0=0)
assert np.array_equal(z, zf[a:b]) and np.array_equal(w, wf[a:b])
# the only exchange: one packed all-reduce of the loss scalars
_, _, s = orc.vmf_kl(kappa[a:b], 4)
tot = allreduce_scalars(s, float(b - a), torch.tensor(w.sum()))
_, _, s_full = orc.vmf_kl(kappa, 4)
assert abs(tot[0].item() - s_full) < 1e-9 * abs(s_full) and tot[1].item() == n
assert abs(tot[2].item() - wf.sum()) < 1e-9
gm = global_mean(s, b - a)
assert abs(gm.item() - s_full / n) < 1e-12
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_rank_gloo_sharding_and_scalar_allreduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT})
    port = 29500 + os.getpid() % 2000
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o


def _run_bench(*flags, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=timeout)
    return out


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no torchrun around it: the parent starts two rank processes, they meet over gloo,
    time between barriers, reduce the maximum, and rank 0's single JSON line comes back with n_gpus 2.  --rehearse
    leaves the GPU out (this machine has none); the GPU suite runs the same launcher with real kernels."""
    import json
    out = _run_bench("--gpus", "2", "--backend", "gloo", "--rows", "4096", "--cpu-rows", "0", "--steps", "3", "--warmup",
                     "1", "--with-allreduce", "--rehearse")
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["rehearsal"] is True and rec["steps"] == 3
    assert len(rec["kernel_ms_per_rank"]) == 2
    assert rec["allreduce"]["count_seen"] == 2 * 4096      # the packed scalars really went through the collective


def test_bench_rehearsal_with_eight_ranks_and_under_a_launcher():
    """the shape of the round-end scaling run, without GPUs: eight ranks started by bench.py itself, and two ranks
    started by `python -m torch.distributed.run` exactly as the driver starts them; every rank reports its own identity
    in the line (on GPUs: ordinal, PCI address, uuid -- here the process ids, which must all differ)."""
    import json
    out = _run_bench("--gpus", "8", "--rows", "1024", "--cpu-rows", "0", "--steps", "2", "--warmup", "1",
                     "--with-allreduce", "--rehearse")
    assert out.returncode == 0, out.stderr
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert rec["n_gpus"] == 8 and rec["ranks_seen"] == 8 and len(rec["kernel_ms_per_rank"]) == 8
    assert [d["rank"] for d in rec["devices"]] == list(range(8))
    assert len({d["pid"] for d in rec["devices"]}) == 8
    assert rec["allreduce"]["count_seen"] == 8 * 1024
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-rows", "0", "--rehearse"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and [d["local_rank"] for d in rec["devices"]] == [0, 1]


def test_bench_fails_when_a_rank_fails():
    out = _run_bench("--gpus", "2", "--rows", "4096", "--cpu-rows", "0", "--steps", "1", "--warmup", "0", "--backend",
                     "no-such-backend")
    assert out.returncode != 0


def test_marker_ranges_load_the_marker_library_only_when_asked():
    """VMF_ROCTX=1: the first entry point (here one that returns on its argument check, no GPU touched) looks the
    marker library up; without the variable nothing of the kind is mapped."""
    code = ("import ctypes, os, sys; sys.path.insert(0, %r)\n"
            "from medicalimageanalysis_amd import _lib\n"
            "rc = _lib.lib().vmf_ive_fwd(0.5, None, None, -1, 0, None)\n"
            "maps = open('/proc/self/maps').read()\n"
            "print(rc, int('rocprofiler-sdk-roctx' in maps))\n") % ROOT
    for flag, want in (("1", "1"), ("0", "0"), (None, "0")):
        env = dict(os.environ)
        env.pop("VMF_ROCTX", None)
        if flag is not None:
            env["VMF_ROCTX"] = flag
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        rc, loaded = out.stdout.split()
        assert int(rc) < 0 and loaded == want, (flag, out.stdout, out.stderr)


def test_allreduce_scalars_argument_checks():
    from medicalimageanalysis_amd import _lib
    lib = _lib.lib()
    buf = (C.c_double * 4)()
    assert lib.vmf_allreduce_scalars(None, buf, 4, None) == -1
    assert lib.vmf_allreduce_scalars(C.c_void_p(1), None, 4, None) == -1
    assert lib.vmf_allreduce_scalars(C.c_void_p(1), buf, 0, None) == -1
    assert lib.vmf_allreduce_scalars(C.c_void_p(1), buf, 65, None) == -1
    assert b"count" in lib.vmf_last_error_string()


def test_streaming_kernels_with_hand_placed_prefetch_use_no_scratch(tmp_path):
    """The fp32 streaming instantiations of vmf_rsample_fwd_kernel (MODE 2) load the next rows from inline asm and wait
    for them by hand: between the two, the compiler believes the destination registers already hold the data.  That is
    only sound while those registers are never spilled (a spilled destination is reused while the load is in flight and
    overwritten when it lands).  Every such instantiation in the built library must have a private segment of zero."""
    import shutil
    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.isdir("/opt/rocm"):
        pytest.skip("no ROCm installation on this machine")
    # a machine that can build the library must also be able to check it: no silent skip
    assert os.path.exists(objdump) and os.path.exists(readelf), "llvm-objdump / llvm-readelf missing from /opt/rocm"
    from medicalimageanalysis_amd import _lib
    so = tmp_path / "libvmf_hip.so"
    shutil.copy(_lib.library_path(), so)
    subprocess.run([objdump, "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path, timeout=300)
    seen = 0
    for f in sorted(tmp_path.iterdir()):
        if "gfx950" not in f.name:
            continue
        notes = subprocess.run([readelf, "--notes", str(f)], check=True, capture_output=True, text=True, timeout=300).stdout
        name = None
        for line in notes.splitlines():
            line = line.strip()
            if line.startswith(".name:"):
                name = line.split(":", 1)[1].strip()
            elif line.startswith(".private_segment_fixed_size:") and name:
                # _Z22vmf_rsample_fwd_kernelIfLi<LPR>ELi<P>ELi2EEv...: T = float, MODE = 2
                if re.match(r"_Z22vmf_rsample_fwd_kernelIfLi\d+ELi\d+ELi2EEv", name):
                    seen += 1
                    assert int(line.split(":")[1]) == 0, (name, line)
                name = None
    assert seen >= 10, seen


def _vregs(text):
    """vector registers named in an operand string: v7 -> {7}, v[4:7] -> {4,5,6,7}"""
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        regs.add(int(a))
    return regs


def test_streaming_kernels_keep_their_hands_off_registers_with_a_load_in_flight(tmp_path):
    """Second half of the hand-placed prefetch's contract: between the `global_load_dwordx4 ... nt` that the kernel issues
    from inline asm and the `s_waitcnt vmcnt(0)` that retires it, NO instruction may read or write the destination
    registers -- the compiler believes they already hold the data and is free to place a copy there (a live-range split,
    a PHI move on the issuing edge); such a copy would read the previous rows' values and corrupt z without any error.
    Checked on the disassembly of every fp32 streaming instantiation (MODE 2) in the built library."""
    import shutil
    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.isdir("/opt/rocm"):
        pytest.skip("no ROCm installation on this machine")
    assert os.path.exists(objdump) and os.path.exists(readelf), "llvm-objdump / llvm-readelf missing from /opt/rocm"
    from medicalimageanalysis_amd import _lib
    so = tmp_path / "libvmf_hip.so"
    shutil.copy(_lib.library_path(), so)
    subprocess.run([objdump, "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path, timeout=300)
    kernels = prefetches = 0
    for f in sorted(tmp_path.iterdir()):
        if "gfx950" not in f.name:
            continue
        syms = subprocess.run([readelf, "--symbols", "--wide", str(f)], check=True, capture_output=True, text=True,
                              timeout=300).stdout
        names = [ln.split()[-1] for ln in syms.splitlines()
                 if " FUNC " in ln and re.search(r"_Z22vmf_rsample_fwd_kernelIfLi\d+ELi\d+ELi2EEv", ln)]
        for name in names:
            dis = subprocess.run([objdump, "-d", f"--disassemble-symbols={name}", str(f)], check=True, capture_output=True,
                                 text=True, timeout=300).stdout
            # (text, address, branch target address or None) per instruction; the scan follows the control flow -- an
            # unconditional branch is taken, a conditional one is followed both ways -- so that a block the code jumps
            # over (the copies of the "no further rows" path right behind a load) is not mistaken for what runs next
            ins = []
            for ln in dis.splitlines():
                mm = re.match(r"^\s+([a-z_0-9]+ .*?)\s*//\s*([0-9A-Fa-f]+):[^<]*(?:<[^>]*\+0x([0-9a-fA-F]+)>)?\s*$", ln)
                if mm:
                    ins.append((mm.group(1).strip(), int(mm.group(2), 16), mm.group(3)))
            assert ins, name
            base = ins[0][1]
            index_of = {addr: i for i, (_, addr, _) in enumerate(ins)}
            kernels += 1
            loads_seen = set()
            seen = set()
            stack = [(0, frozenset())]
            while stack:
                i, in_flight = stack.pop()
                while i < len(ins):
                    if (i, in_flight) in seen:
                        break
                    seen.add((i, in_flight))
                    text, _, target = ins[i]
                    op, _, operands = text.partition(" ")
                    if op == "global_load_dwordx4" and operands.rstrip().endswith(" nt"):
                        dst = _vregs(operands.split(",")[0])
                        assert not (dst & in_flight), (name, i, text)
                        # the address registers are read by this very instruction: they must not be in flight either
                        assert not (_vregs(",".join(operands.split(",")[1:])) & in_flight), (name, i, text)
                        in_flight = in_flight | dst
                        loads_seen.add(i)
                    elif op == "s_waitcnt" and "vmcnt(0)" in operands:
                        in_flight = frozenset()
                    elif in_flight and op.startswith(("v_", "global_", "ds_", "buffer_", "scratch_", "flat_")):
                        touched = _vregs(operands) & in_flight
                        assert not touched, (name, i, text, sorted(touched), [t for t, _, _ in ins[max(0, i - 3):i + 1]])
                    if op == "s_endpgm":
                        assert not in_flight, (name, "a prefetch is never waited for")
                        break
                    if op == "s_branch" or op.startswith("s_cbranch"):
                        assert target is not None, (name, i, text)
                        j = index_of[base + int(target, 16)]
                        if op == "s_branch":
                            i = j
                            continue
                        stack.append((j, in_flight))
                    i += 1
            prefetches += len(loads_seen)
    assert kernels >= 10 and prefetches >= 2 * kernels, (kernels, prefetches)
