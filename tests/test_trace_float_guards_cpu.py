"""-DKLANG_GPU_TRACE_FLOAT renames `float` to the tracing signal in a patch's own text (include/klang/klang.h).  What that must never do silently — change
sizeof(float), let a signal go through the C library's byte copiers — is a COMPILE ERROR with a message; ordinary patch text still compiles.  Syntax-only compiles
against the facade header: no GPU, no link."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXX = "/opt/rocm/lib/llvm/bin/clang++" if os.path.exists("/opt/rocm/lib/llvm/bin/clang++") else "g++"
HEAD = '#include <klang.h>\nusing namespace klang::optimised;\n'
FX = "struct P : Effect { %s void process() { %s } };\n"


def compiles(body_decl, body, trace=True):
    src = HEAD + FX % (body_decl, body)
    cmd = [CXX, "-std=c++17", "-fsyntax-only", "-x", "c++", "-", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "include", "klang")] + (["-DKLANG_GPU_TRACE_FLOAT"] if trace else [])
    r = subprocess.run(cmd, input=src, capture_output=True, text=True)
    return r.returncode == 0, r.stderr


def test_ordinary_float_code_compiles_under_the_switch():
    ok, err = compiles("static float clip(float x) { if (x > 1) return 1; if (x < -1) return -1; return x; }", "clip(in * 2) >> out;")
    assert ok, err[-1500:]
    ok, err = compiles("", "unsigned n = sizeof(int) + sizeof(double); (void)n; in >> out;")          # sizeof of other types is untouched
    assert ok, err[-1500:]


@pytest.mark.parametrize("decl,body,needle", [
    ("", "float t = in; unsigned n = sizeof(float); (void)n; (void)t; in >> out;", "sizeof(float) in a patch compiled with -DKLANG_GPU_TRACE_FLOAT"),
    ("", "float buf[4]; unsigned n = sizeof(buf); (void)n; in >> out;", "sizeof(float) in a patch compiled with -DKLANG_GPU_TRACE_FLOAT"),
    ("", "float a[4], b[4]; memcpy(a, b, 16); in >> out;", "deleted"),
    ("", "float a[4]; std::memset(a, 0, 16); in >> out;", "deleted"),
    ("", "union { float f; unsigned u; } x; x.f = 1; (void)x; in >> out;", "error"),
])
def test_byte_level_uses_of_the_traced_float_do_not_compile(decl, body, needle):
    ok, err = compiles(decl, body)
    assert not ok and needle in err, err[-1500:]


def test_byte_copies_of_a_signal_do_not_compile_without_the_switch_either():
    ok, err = compiles("signal s[2];", "memcpy(&s[0], &s[1], sizeof(signal)); in >> out;", trace=False)
    assert not ok and "deleted" in err, err[-1500:]
    ok, err = compiles("", "float a[4], b[4]; memcpy(a, b, sizeof(a)); (void)a; in >> out;", trace=False)    # plain floats without the switch: plain C
    assert ok, err[-1500:]
