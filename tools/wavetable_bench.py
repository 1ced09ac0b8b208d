#!/usr/bin/env python3
"""tools/wavetable_bench.py — gather rate of the wavetable node on one MI355X.
A graph bank whose process() is `wavetable >> out`: V voices with random increments, either all reading ONE 2048-sample table
(what klg_table_upload's content dedup gives a bank of notes built from the same oscillator) or `--tables T` distinct ones
(voice v reads table v % T).  Prints one JSON line per case: voice*samples/s and the kernel time per 256-sample block."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import klang_amd  # noqa: E402

PROGRAM = "klgg 1\nctl 0\nnode 0 wavetable\nop osc 0 -1 -1 0 0\nret 0\nend\n"


def run(V, T, size=2048, N=256, steps=50, warmup=5):
    P = 128
    bank = klang_amd.SynthBank(PROGRAM, synths=V // P, notes=P, max_block=N)
    rng = np.random.default_rng(5)
    ids = [bank.table_upload(np.sin(2 * np.pi * (np.arange(size) + k) / size).astype(np.float32), dedup=False) for k in range(T)]
    words = np.zeros((V, 6), np.uint32)
    words[:, 0] = 1
    words[:, 1] = rng.uniform(1.0, 40.0, V).astype(np.float32).view(np.uint32)           # increment (55 Hz .. 2 kHz at 48 kHz)
    words[:, 2] = rng.uniform(0, size - 1, V).astype(np.float32).view(np.uint32)
    words[:, 5] = np.asarray(ids, np.uint32)[np.arange(V) % T]
    for c0 in range(0, V, 1 << 16):
        bank.voices_upload(np.arange(c0, min(V, c0 + (1 << 16)), dtype=np.int32), words[c0:c0 + (1 << 16)])
    mix = torch.zeros((2, N), dtype=torch.float32, device="cuda")
    torch.cuda.set_stream(torch.cuda.Stream())      # a stream of our own: handle 0 (torch's default) means "the bank's own stream" to the library, which torch's clears are not ordered with
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(warmup):
        mix.zero_(); bank.process_device(mix.data_ptr(), N, st)
    torch.cuda.synchronize(); bank.timing_begin(); t0 = time.perf_counter()
    for _ in range(steps):
        mix.zero_(); bank.process_device(mix.data_ptr(), N, st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n, ms = bank.timing_end()
    bank.close()
    return dict(voices=V, tables=T, table_bytes=size * 4, value=V * N * steps / dt, unit="voice*samples/s", kernel_ms=ms / n,
                gather_GBps=V * N * 8 / (ms / n * 1e-3) / 1e9, mix_abs_sum=float(mix.abs().sum().item()))


if __name__ == "__main__":
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    for T in (1, 64, 4096, 65536):
        print(json.dumps(run(V, T)), flush=True)
