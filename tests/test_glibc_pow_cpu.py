"""klang_amd/csrc/klg_glibc_pow.hpp (glibc 2.35's double pow / exp2 restated: graph `func 1 / 2`, Modular.k) compiled for the HOST against the host's libm: a subset of
tools/verify_glibc_pow.cpp's full run (all 2^32 floats + 10^9 random pairs: 0 differ).  Needs a host with FMA — the C library then runs `__pow_fma`, the variant the header
restates (the fixtures were generated, and the GPU boxes run, on such hosts)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_restated_pow_and_exp2_equal_libm(tmp_path):
    if " fma " not in open("/proc/cpuinfo").read().replace("\n", " ") + " ":
        pytest.skip("no FMA on this host: its libm runs another variant of pow")
    exe = str(tmp_path / "verify_glibc_pow")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-fopenmp", os.path.join(ROOT, "tools", "verify_glibc_pow.cpp"), "-o", exe, "-lm"], check=True)
    r = subprocess.run([exe, "2000000", "251"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "pow(10, f) 0 differ, pow(2, f) 0 differ, exp2(f) 0 differ" in r.stdout and "2000000 random (x, y): 0 differ" in r.stdout, r.stdout


def test_the_committed_tables_are_the_pinned_c_librarys(tmp_path):
    """klg_glibc_tables.hpp is what tools/extract_glibc_tables.py reads out of this image's libm.so.6 (glibc 2.35)."""
    import sys
    libm = "/lib/x86_64-linux-gnu/libm.so.6"
    if not os.path.exists(libm):
        pytest.skip("no x86-64 glibc here")
    out = str(tmp_path / "tables.hpp")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "extract_glibc_tables.py"), libm, out], capture_output=True, text=True)
    if r.returncode != 0 and "table not found" in r.stderr:
        pytest.skip("another glibc: " + r.stderr.strip().splitlines()[-1])
    assert r.returncode == 0, r.stderr
    assert open(out).read() == open(os.path.join(ROOT, "klang_amd", "csrc", "klg_glibc_tables.hpp")).read()
