"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer: a standalone driver (oracle/selftest.c) runs
its entry points on small ragged inputs.  The parity tests lean on the oracle, so the oracle itself must be clean."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_and_ubsan():
    if shutil.which("gcc") is None and shutil.which("cc") is None:
        pytest.skip("no C compiler")
    build = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "selftest"], capture_output=True, text=True)
    if build.returncode != 0:
        pytest.skip("sanitizer runtime not available: " + build.stderr[-300:])
    env = dict(os.environ, OMP_NUM_THREADS="2", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    run = subprocess.run([os.path.join(ROOT, "oracle", "selftest")], capture_output=True, text=True, env=env,
                         timeout=300)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "0 failure(s)" in run.stdout
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr
