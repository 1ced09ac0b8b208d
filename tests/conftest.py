"""pytest configuration: markers, paths, on-demand builds, and an honest account of what could not run.

The GPU suite is self-sufficient for everything that can be built where it runs: the product library (hipcc), the oracle (gcc) and the DSL
facade drivers over OUR patches (tests/cpp/_bin) are built on demand.  Drivers that contain the REFERENCE's .k files (oracle/_ref/facade_*,
built by `make -C tests/cpp shipped` where /root/reference exists) cannot be rebuilt on a GPU box: they travel with the tree.  A test that
needs one and does not find it is skipped with a marked reason — and the session then FAILS with a summary of how many tests that was, so
a silent 40-test skip can never pass for green (set KLG_ALLOW_MISSING_REF=1 to accept a reduced run knowingly)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE = os.path.join(ROOT, "oracle")
NEEDS_REF = "built only where the reference's .k files exist"      # the skip reason every test that needs an oracle/_ref binary uses
_missing = []


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Before the first GPU test: build what can be built here (the product .so, the facade drivers over our own patches)."""
    if not any("gpu" in it.keywords for it in items):
        return
    lib = os.path.join(ROOT, "klang_amd", "libklang_mi355.so")
    if not os.path.exists(lib):
        subprocess.run(["bash", os.path.join(ROOT, "klang_amd", "csrc", "build.sh")], check=True)
    if not os.path.isdir(os.path.join(ROOT, "tests", "cpp", "_bin")) or not os.listdir(os.path.join(ROOT, "tests", "cpp", "_bin")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True)
    if os.path.isdir("/root/reference") and not os.path.exists(os.path.join(ORACLE, "_ref", "facade_fx_gain")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "shipped"], check=True)


def pytest_runtest_logreport(report):
    if report.skipped and NEEDS_REF in str(report.longrepr):
        _missing.append(report.nodeid)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if _missing:
        terminalreporter.section("tests that need prebuilt oracle/_ref binaries")
        terminalreporter.write_line(f"{len(_missing)} tests were SKIPPED because their driver (a reference .k file compiled against the facade) is not in oracle/_ref/:")
        for n in _missing[:12]:
            terminalreporter.write_line("  " + n)
        if len(_missing) > 12:
            terminalreporter.write_line(f"  ... and {len(_missing) - 12} more")
        terminalreporter.write_line("build them where /root/reference exists (python -c 'import __graft_entry__ as g; g.build()') and let them travel with the tree.")


def pytest_sessionfinish(session, exitstatus):
    if _missing and os.environ.get("KLG_ALLOW_MISSING_REF") != "1" and session.exitstatus == 0:
        session.exitstatus = 1


@pytest.fixture(scope="session")
def oracle_build():
    """Builds oracle/_build (C restatement; gcc only, no reference needed) and returns its directory."""
    subprocess.run(["make", "-C", ORACLE, "oracle"], check=True, stdout=subprocess.DEVNULL)
    return os.path.join(ORACLE, "_build")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.hookimpl(wrapper=True)
def pytest_runtest_call(item):
    """A/B reference kernels (the one-wave PingPong, SuperSaw's oscillator-per-lane and pair-per-lane forms) are in the library only when it is built with
    klang_amd/csrc/build.sh -DKLG_AB_KERNELS: a test that asks for one through its environment switch is skipped on the shipped build (the library says so itself)."""
    try:
        return (yield)
    except Exception as e:   # noqa: BLE001
        if "built without -DKLG_AB_KERNELS" in str(e):
            pytest.skip("A/B reference kernel: the library was built without -DKLG_AB_KERNELS")
        raise
