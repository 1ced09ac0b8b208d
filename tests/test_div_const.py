"""klg::div_const (klang_amd/csrc/klg_device.hpp): the three-operation constant division is only used for divisors checked
against IEEE x / y on every float bit pattern by tools/verify_div_const.c.  This runs that check on a strided subset (the full
2^32 sweep is `verify_div_const 1 ...`, ~20 s per divisor) and shows the check has teeth: 6 is NOT exact and is not accepted."""
import os, re, subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ACCEPTED = {"7": 0x40e00000, "3": 0x40400000, "5": 0x40a00000, "9": 0x41100000, "2.5": 0x40200000}


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("divc") / "verify_div_const")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fopenmp", os.path.join(ROOT, "tools", "verify_div_const.c"), "-o", exe, "-lm"])
    return exe


def test_accepted_divisors_are_exact(checker):
    out = subprocess.run([checker, "61"] + list(ACCEPTED), capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert [l.split()[1] for l in out.stdout.splitlines()] == ["0"] * len(ACCEPTED)


def test_checker_rejects_inexact_divisor(checker):
    out = subprocess.run([checker, "61", "6"], capture_output=True, text=True)
    assert out.returncode == 1 and int(out.stdout.split()[1]) > 0


def test_device_and_codegen_lists_match_the_checked_set():
    dev = open(os.path.join(ROOT, "klang_amd", "csrc", "klg_device.hpp")).read()
    gen = open(os.path.join(ROOT, "klang_amd", "csrc", "klg_graph.hpp")).read()
    dev_set = set(int(x, 16) for x in re.findall(r"YBITS == (0x[0-9a-f]{8})u", dev))
    gen_line = next(l for l in gen.splitlines() if "div_const<" in l and "y ==" in l)
    gen_set = set(int(x, 16) for x in re.findall(r"y == (0x[0-9a-f]{8})u", gen_line))
    assert dev_set == gen_set == set(ACCEPTED.values())
