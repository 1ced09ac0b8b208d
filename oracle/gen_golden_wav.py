#!/usr/bin/env python3
"""oracle/gen_golden_wav.py — TEST INFRASTRUCTURE.  Writes four small mono WAV files (PCM 8 / 16 / 32, float 32) under tests/golden/wav/ and
decodes each with the GENUINE reference header's File::WAV (oracle/_ref/ref_wav, built by `make -C oracle ref` where /root/reference
exists); the decoded floats go to tests/golden/wav_expected.npz.  tests/test_host_render.py compares include/klang/host/wav.hpp with them."""
import os
import struct
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.environ.get("KLG_GOLDEN_OUT") or os.path.join(HERE, "..", "tests", "golden")
OUT = os.path.join(GOLD, "wav")
os.makedirs(OUT, exist_ok=True)


def riff(path, fmt, bits, payload, rate=48000, ch=1):
    block = ch * bits // 8
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(payload)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, fmt, ch, rate, rate * block, block, bits))
        f.write(b"data" + struct.pack("<I", len(payload)) + payload)


rng = np.random.default_rng(7)
n = 257
files = {
    "u8": (1, 8, np.concatenate([[0, 1, 127, 128, 129, 254, 255], rng.integers(0, 256, n - 7)]).astype(np.uint8).tobytes()),
    "s16": (1, 16, np.concatenate([[-32768, -1, 0, 1, 32767], rng.integers(-32768, 32768, n - 5)]).astype("<i2").tobytes()),
    "s32": (1, 32, np.concatenate([[-2**31, -1, 0, 1, 2**31 - 1, 16777217, -16777217], rng.integers(-2**31, 2**31, n - 7)]).astype("<i4").tobytes()),
    "f32": (3, 32, np.concatenate([[0.0, -0.0, 1.0, -1.0, 1e-40, 0.333333343], rng.uniform(-1, 1, n - 6)]).astype("<f4").tobytes()),
}
expected = {}
for name, (fmt, bits, payload) in files.items():
    path = os.path.join(OUT, name + ".wav")
    riff(path, fmt, bits, payload)
    lines = subprocess.run([os.path.join(HERE, "_ref", "ref_wav"), path], check=True, capture_output=True, text=True).stdout.split()
    count = int(lines[0])
    expected[name] = np.array([int(x, 16) for x in lines[1:1 + count]], dtype=np.uint32)
    print(name, count, "samples")
np.savez(os.path.join(OUT, "..", "wav_expected.npz"), **expected)
