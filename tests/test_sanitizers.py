"""the reference gets its memory safety from Rust; the C++ on the CPU side gets it from one run under
AddressSanitizer + UndefinedBehaviorSanitizer.  Both builds are kept as make targets:
  make -C raytracing_amd/csrc asan   host_scene.cpp + tools/host_scene_san_main.cpp -> raytracing_amd/host_scene_asan
  make -C oracle asan                oracle_capi.cpp -> oracle/liboracle_asan.so
The oracle build is exercised by running its own known-answer tests in a child interpreter with the sanitizer
runtime preloaded (a sanitized .so cannot be dlopen'ed into an unsanitized python otherwise)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make(directory, target):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, directory), target], stdout=subprocess.DEVNULL)


def test_host_scene_under_asan_ubsan():
    _make("raytracing_amd/csrc", "asan")
    out = subprocess.run([os.path.join(ROOT, "raytracing_amd", "host_scene_asan")], capture_output=True, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "sanitizer driver ok" in out.stdout and "runtime error" not in out.stderr


def test_oracle_under_asan_ubsan():
    if os.environ.get("ORACLE_LIB"):
        pytest.skip("already inside the sanitized child run")
    _make("oracle", "asan")
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, ORACLE_LIB=os.path.join(ROOT, "oracle", "liboracle_asan.so"), LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")  # python itself "leaks" at exit
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_kats.py"), "-x", "-q",
                          "-m", "not gpu", "-p", "no:cacheprovider"], capture_output=True, text=True, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]
