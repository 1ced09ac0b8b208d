#!/usr/bin/env python3
"""tools/pluck_bench.py — delay lines in notes at scale: the recorded tests/patches/pluck.k (tests/golden/pluck_recorded.klgg: a
Karplus-Strong string, Delay<4800> per voice, read head + fractional tap + write per sample) on V voices with random pitches and
unaligned write cursors (every voice started at another time).  Prints voice*samples/s and the ring bytes moved per second
(algorithmic: 5 floats per voice*sample)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import klang_amd  # noqa: E402

PROGRAM = open(os.path.join(ROOT, "tests", "golden", "pluck_recorded.klgg")).read()
SIZE = 4800


def main():
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
    N, P = 256, 128
    f32 = np.float32
    rng = np.random.default_rng(3)
    bank = klang_amd.SynthBank(PROGRAM, synths=V // P, notes=P, max_block=N)
    W = bank.state_bytes // 4
    assert W == 1 + 15 + 4 + 3 + 9 + 1, W
    words = np.zeros((V, W), np.uint32)
    freq = 440.0 * 2.0 ** ((rng.integers(36, 97, V) - 69) / 12.0)
    t = (48000.0 / freq - 2).astype(f32)
    pos = rng.integers(0, SIZE, V)
    read = (pos - 1).astype(f32) - t
    read = np.where(read < 0, read + f32(SIZE), read).astype(f32)
    lastpos = read.astype(np.int32)
    words[:, 0] = 1                                                    # Sustain
    words[:, 1 + 4] = 2                                                # excitation envelope: finished (stage Off)
    d0 = 1 + 15
    words[:, d0 + 0] = pos; words[:, d0 + 1] = lastpos; words[:, d0 + 2] = (read - lastpos.astype(f32)).astype(f32).view(np.uint32); words[:, d0 + 3] = t.view(np.uint32)
    i0 = d0 + 4
    words[:, i0 + 0] = f32(0.15).view(np.uint32); words[:, i0 + 1] = f32(0.85).view(np.uint32)
    a0 = i0 + 3
    words[:, a0 + 0] = f32(1).view(np.uint32); words[:, a0 + 1] = f32(1).view(np.uint32); words[:, a0 + 4] = (0 | (2 << 2)); words[:, a0 + 7] = f32(1).view(np.uint32); words[:, a0 + 8] = f32(0.5).view(np.uint32)
    words[:, a0 + 9] = (t * f32(2.0 / 3.0)).astype(f32).view(np.uint32)
    for c0 in range(0, V, 1 << 16):
        bank.voices_upload(np.arange(c0, min(V, c0 + (1 << 16)), dtype=np.int32), words[c0:c0 + (1 << 16)])
    mix = torch.zeros((2, N), dtype=torch.float32, device="cuda")
    torch.cuda.set_stream(torch.cuda.Stream())      # a stream of our own: handle 0 (torch's default) means "the bank's own stream" to the library, which torch's clears are not ordered with
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        mix.zero_(); bank.process_device(mix.data_ptr(), N, st)
    torch.cuda.synchronize(); bank.timing_begin(); t0 = time.perf_counter()
    steps = 40
    for _ in range(steps):
        mix.zero_(); bank.process_device(mix.data_ptr(), N, st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n, ms = bank.timing_end()
    print(json.dumps(dict(voices=V, delay_size=SIZE, lines_GB=V * SIZE * 4 / 1e9, value=V * N * steps / dt, unit="voice*samples/s", kernel_ms=ms / n,
                          ring_GBps_algorithmic=V * N * 20 / (ms / n * 1e-3) / 1e9)))
    bank.close()


if __name__ == "__main__":
    main()
