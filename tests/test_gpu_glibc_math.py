"""The C library's double pow (constant base) / exp2 on the device (klang_amd/csrc/klg_glibc_pow.hpp; graph OP_FUNC 1 / 2), klang's power() with a literal exponent (OP_POWC) and
the float-through-int conversion (OP_TRUNC): tiny `kind effect` programs `in -> op -> out`, fed floats of every exponent range and the special values, against the HOST's libm
called through ctypes — bit for bit.  (tools/verify_glibc_pow.cpp checks the same header, compiled for the host, on all 2^32 floats and 10^9 random pairs.)"""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

K, N = 64, 256


def inputs(seed):
    rng = np.random.default_rng(seed)
    n = K * N
    e = rng.integers(1, 255, n).astype(np.uint32)                                    # every normal exponent ...
    e[: n // 2] = rng.integers(100, 140, n // 2).astype(np.uint32)                   # ... half of them where the results are finite floats
    x = ((rng.integers(0, 2, n).astype(np.uint32) << 31) | (e << 23) | rng.integers(0, 1 << 23, n).astype(np.uint32)).view(np.float32)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.0, -1.0, 2.0, 10.0, 38.5, -37.9, -45.0, 127.99, 128.0, -126.0, -149.0, -150.5, 1023.5, 1024.0, -1074.0, -1075.0,
                        2147483648.0, -2147483648.0, 2147483520.0, 3e9, -3e9, 0.5, -0.5, 0.99999994, 19999.99, 20000.0], np.float32)
    x[: len(special)] = special
    return x.reshape(K, 1, N)


def run(program, x):
    import torch
    import klang_amd
    bank = klang_amd.FxBank(program, K, max_block=N, channels=1)
    io = torch.from_numpy(x.copy()).cuda()
    bank.process_device(io.data_ptr(), N, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = io.cpu().numpy()
    bank.close()
    return out


def same(got, want):
    g, w = got.view(np.uint32).ravel(), want.view(np.uint32).ravel()
    nan = np.isnan(got.ravel()) & np.isnan(want.ravel())
    return np.flatnonzero((g != w) & ~nan)


def func_program(imm):
    return f"klgg 1\nkind effect 1\nctl 0\nop in 0 -1 -1 -1 0\nop f2d 1 0 -1 -1 0\nop func 2 1 -1 -1 {imm:08x}\nop d2f 3 2 -1 -1 0\nret 3\nend\n"


@pytest.mark.parametrize("name,imm,base", [("pow10", 0x41200002, 10.0), ("pow2", 0x40000002, 2.0), ("pow_half", 0x3F000002, 0.5), ("pow_1e-3ish", 0x3A830002, None), ("exp2", 1, None)])
def test_double_pow_and_exp2_equal_the_c_library(name, imm, base):
    libm = ctypes.CDLL("libm.so.6")
    libm.pow.restype = ctypes.c_double; libm.pow.argtypes = [ctypes.c_double, ctypes.c_double]
    libm.exp2.restype = ctypes.c_double; libm.exp2.argtypes = [ctypes.c_double]
    if imm & 0xFF == 2 and base is None:
        base = float(np.array([imm & 0xFFFFFF00], np.uint32).view(np.float32)[0])
    x = inputs(11 + (imm & 0xFF))
    got = run(func_program(imm), x)
    want = np.array([np.float32(libm.exp2(float(v)) if imm == 1 else libm.pow(base, float(v))) for v in x.ravel()], np.float32).reshape(x.shape)
    bad = same(got, want)
    assert len(bad) == 0, f"{name}: {len(bad)} of {x.size} differ, first input {x.ravel()[bad[0]]!r}: device {got.ravel()[bad[0]]!r}, libm {want.ravel()[bad[0]]!r}"
    assert np.isfinite(got).sum() > x.size // 8


@pytest.mark.parametrize("e", [0.0, 1.0, 2.0, 3.0, 4.0, -1.0, -2.0, -3.0, -4.0])
def test_power_with_a_literal_exponent(e):
    """klang.h:188-218 on a float base: base == 10 first ((float)exp(e * ln 10), the C library's, in double on the float product), then the written-out products."""
    libm = ctypes.CDLL("libm.so.6")
    libm.exp.restype = ctypes.c_double; libm.exp.argtypes = [ctypes.c_double]
    bits = int(np.array([e], np.float32).view(np.uint32)[0])
    x = inputs(5)
    got = run(f"klgg 1\nkind effect 1\nctl 0\nop in 0 -1 -1 -1 0\nop powc 1 0 -1 -1 {bits:08x}\nret 1\nend\n", x)
    f = x.astype(np.float32)
    with np.errstate(all="ignore"):
        m = int(abs(e))
        p = np.ones_like(f) if m == 0 else f.copy()
        for _ in range(1, m):
            p = (p * f).astype(np.float32)
        if e < 0:
            p = (np.float32(1.0) / p).astype(np.float32)
    ten = np.float32(libm.exp(float(np.float32(e) * np.float32(2.3025850929940456840179914546843642076011014886287729760333279009))))
    want = np.where(f == np.float32(10.0), ten, p).astype(np.float32)
    bad = same(got, want)
    assert len(bad) == 0, f"exponent {e}: {len(bad)} differ, first input {x.ravel()[bad[0]]!r}: device {got.ravel()[bad[0]]!r}, expected {want.ravel()[bad[0]]!r}"


def test_float_through_int():
    x = inputs(9)
    got = run("klgg 1\nkind effect 1\nctl 0\nop in 0 -1 -1 -1 0\nop trunc 1 0 -1 -1 0\nret 1\nend\n", x)
    f = x.ravel().astype(np.float64)
    with np.errstate(all="ignore"):
        want = np.where(np.isnan(f) | (np.abs(f) >= 2147483648.0), -2147483648.0, np.trunc(f)).astype(np.float32).reshape(x.shape)
    want = np.where(want == 0, np.float32(0.0), want)                                # (an int has no negative zero)
    bad = same(got, want)
    assert len(bad) == 0, f"{len(bad)} differ, first input {x.ravel()[bad[0]]!r}: device {got.ravel()[bad[0]]!r}, expected {want.ravel()[bad[0]]!r}"
