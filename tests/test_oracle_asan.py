"""CPU: the C oracle under AddressSanitizer + UBSan (`make -C oracle asan`) passes its golden cases -- the checker the
GPU parity tests lean on has no out-of-bounds access or undefined behaviour on the fixtures (sanitizers
run on the CPU build only).  The sanitizer runtime must be the first library of the process, so the cases run in a child
interpreter with it preloaded."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_golden_cases_under_asan_ubsan():
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "oracle", "_asan", "liboracle_asan.so")
    assert os.path.exists(lib)
    pre = [subprocess.run([gcc, "-print-file-name=" + n], check=True, capture_output=True, text=True).stdout.strip()
           for n in ("libasan.so", "libubsan.so")]
    pre = [p for p in pre if os.path.isabs(p) and os.path.exists(p)]
    if not pre:
        pytest.skip("no sanitizer runtime next to gcc")
    env = dict(os.environ, LD_PRELOAD=":".join(pre), CLOUDAAE_ORACLE_LIB=lib,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23", UBSAN_OPTIONS="halt_on_error=1:exitcode=24",
               OMP_NUM_THREADS="4")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q",
                          "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, cwd=ROOT, timeout=900)
    text = out.stdout + out.stderr
    assert "AddressSanitizer" not in text and "runtime error:" not in text, text[-4000:]
    assert out.returncode == 0, text[-4000:]
    assert " passed" in out.stdout
