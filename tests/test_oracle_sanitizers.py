"""The oracle's C sources under AddressSanitizer + UndefinedBehaviorSanitizer.  CPU only: the GPU pool
refuses sanitizer runs.  `make -C oracle asan` builds libjsdr_oracle_asan.so; a child Python with the sanitizer runtime
preloaded runs the oracle's own test files against it (JSDR_ORACLE_SO makes tests/oracle_lib.py load that build).  Any
report -- heap / stack / global overflow, use after free, signed overflow, misaligned access, shift out of range -- ends the
child with a non-zero status."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
# the files that exercise the oracle alone (no libjsdr_hip.so, no GPU)
FILES = ["test_oracle_kat.py", "test_oracle_tables.py", "test_reference_fixtures.py", "test_fir_decimate_oracle.py",
         "test_demod.py", "test_formats.py"]


def runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_suite_is_clean_under_asan_and_ubsan():
    asan = runtime("libasan.so")
    if asan is None:
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-s", "-C", ODIR, "asan"])
    so = os.path.join(ODIR, "libjsdr_oracle_asan.so")
    env = dict(os.environ, JSDR_ORACLE_SO=so, LD_PRELOAD=asan,
               # CPython itself leaks by design at exit; the oracle's own allocations are checked by the suite's frees
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "-m", "not gpu"] +
                       [os.path.join(ROOT, "tests", f) for f in FILES], env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout[-3000:] + r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
