#!/usr/bin/env python3
"""tools/host_path_bench.py — the PCIe-inclusive rates of the host-buffer entry points (bench.py's `value` is the device-resident
rate): klg_process (mix + note stages come back every block) and klg_fx_process (the whole [K][2][n] block goes both ways)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import klang_amd
N = 256
out = []
for V in (1 << 16, 1 << 20):
    bank = klang_amd.SynthBank("sub2a", synths=V // 128, notes=128, max_block=N)
    rng = np.random.default_rng(1)
    bank.note_on_many(np.arange(V) // 128, rng.integers(36, 97, V), np.full(V, 0.8, np.float32))
    mix = np.zeros((2, N), np.float32)
    for _ in range(30): bank.process(mix)
    t0 = time.perf_counter(); steps = 50
    for _ in range(steps): bank.process(mix)
    dt = time.perf_counter() - t0
    out.append(dict(path="klg_process (host buffers)", voices=V, ms_per_block=1e3 * dt / steps, voice_samples_per_s=V * N * steps / dt, d2h_bytes_per_block=2 * N * 4))
    bank.close()
for K in (256, 4096):
    bank = klang_amd.FxBank("pingpong", K, max_block=N)
    io = (np.random.default_rng(2).uniform(-.5, .5, (K, 2, N))).astype(np.float32)
    for _ in range(5): bank.process(io)
    t0 = time.perf_counter(); steps = 30
    for _ in range(steps): bank.process(io)
    dt = time.perf_counter() - t0
    out.append(dict(path="klg_fx_process (host buffers)", instances=K, ms_per_block=1e3 * dt / steps, inst_samples_per_s=K * N * steps / dt, pcie_bytes_per_block=2 * K * 2 * N * 4))
    bank.close()
for r in out: print(json.dumps(r))
