#!/usr/bin/env python3
"""tools/bench_all.py — measures every BASELINE.json config on ONE MI355X and the oracle (CPU, 1 core) beside it.
Writes one JSON object per config to stdout / --out.  Not the driver's bench (that is bench.py); this is the
table DESIGN.md quotes.  Inputs are resident in HBM when the timed region starts."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STORE_WORDS = {"sub2a": 8, "sub2b": 19, "supersaw": 12, "fm3": 20, "fm4": 25, "sine": 2, "bsine": 2}
FX_BYTES_PER_SAMPLE = {"pingpong": 32, "reverb": 312}     # SURVEY.md §8(d) algorithmic bytes per instance*sample


def load_oracle():
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True, stdout=subprocess.DEVNULL)
    ko = C.CDLL(os.path.join(ROOT, "oracle", "_build", "libklang_oracle.so"))
    ko.ko_bank_create.restype = C.c_void_p
    ko.ko_bank_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float]
    ko.ko_bank_note_on.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_long]
    ko.ko_bank_note_off.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float]
    ko.ko_bank_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    ko.ko_patch_from_name.argtypes = [C.c_char_p]
    ko.ko_fxbank_create.restype = C.c_void_p
    ko.ko_fxbank_create.argtypes = [C.c_int, C.c_int, C.c_float]
    ko.ko_fxbank_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    return ko


def cpu_synth(ko, patch, N, budget):
    notes = 128 if patch in ("sub2a", "sine", "bsine") else 32
    synths = 128 // notes
    bank = ko.ko_bank_create(ko.ko_patch_from_name(patch.encode()), synths, notes, C.c_float(48000.0))
    rng = np.random.default_rng(1)
    for sy in range(synths):
        for p in rng.integers(36, 97, size=notes):
            ko.ko_bank_note_on(bank, sy, int(p), C.c_float(0.8), 1)
    mix = np.zeros((2, N), np.float32)
    t0 = time.perf_counter(); blocks = 0
    while time.perf_counter() - t0 < budget:
        for _ in range(20):
            ko.ko_bank_process(bank, None, mix.ctypes.data_as(C.c_void_p), None, N)
        blocks += 20
    return 128 * N * blocks / (time.perf_counter() - t0)


def cpu_fx(ko, patch, N, budget):
    K = 4
    bank = ko.ko_fxbank_create(ko.ko_patch_from_name(patch.encode()), K, C.c_float(48000.0))
    io = (np.random.default_rng(2).uniform(-.5, .5, size=(K, 2, N))).astype(np.float32)
    t0 = time.perf_counter(); blocks = 0
    while time.perf_counter() - t0 < budget:
        for _ in range(5):
            ko.ko_fxbank_process(bank, io.ctypes.data_as(C.c_void_p), N)
        blocks += 5
    return K * N * blocks / (time.perf_counter() - t0)


def gpu_synth(patch, voices, N, steps, warmup, release=False):
    import torch
    import klang_amd
    notes = 128 if patch in ("sub2a", "sine", "bsine") else 32
    synths = max(1, voices // notes)
    bank = klang_amd.SynthBank(patch, synths=synths, notes=notes, max_block=N)
    rng = np.random.default_rng(20250314)
    pitches = rng.integers(36, 97, size=bank.voices)
    for v in range(bank.voices):
        bank.random(v + 1)
        bank.note_on(v // notes, int(pitches[v]), 0.8)
    mix = torch.zeros((2, N), dtype=torch.float32, device="cuda")
    torch.cuda.set_stream(torch.cuda.Stream())      # a stream of our own: handle 0 (torch's default) means "the bank's own stream" to the library, which torch's clears are not ordered with
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(warmup):
        mix.zero_(); bank.process_device(mix.data_ptr(), N, stream)
    if release:
        for v in range(bank.voices):
            bank.note_off(v // notes, int(pitches[v]), 0.0)
        mix.zero_(); bank.process_device(mix.data_ptr(), N, stream)
    torch.cuda.synchronize()
    bank.timing_begin()
    t0 = time.perf_counter()
    for _ in range(steps):
        mix.zero_(); bank.process_device(mix.data_ptr(), N, stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    launches, kms = bank.timing_end()
    alive = int((bank.stages() != 3).sum())
    res = dict(patch=patch, voices=bank.voices, block=N, steps=steps, value=bank.voices * N * steps / dt, ms_per_step=1e3 * dt / steps,
               kernel_ms=kms / launches, state_bytes=bank.state_bytes,
               alg_bytes_per_launch=bank.voices * (bank.state_bytes + 4 * STORE_WORDS[patch]) + 2 * N * 4, voices_alive_after=alive,
               mix_abs_sum=float(mix.abs().sum().item()))
    res["hbm_gbs"] = res["alg_bytes_per_launch"] / (1e-3 * res["kernel_ms"]) / 1e9
    bank.close()
    return res


def gpu_fx(patch, K, N, steps, warmup):
    import torch
    import klang_amd
    bank = klang_amd.FxBank(patch, K, max_block=N)
    g = torch.Generator(device="cuda").manual_seed(1)
    io = (torch.rand((K, 2, N), device="cuda", generator=g) - 0.5)
    torch.cuda.set_stream(torch.cuda.Stream())      # a stream of our own: handle 0 (torch's default) means "the bank's own stream" to the library, which torch's clears are not ordered with
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(warmup):
        bank.process_device(io.data_ptr(), N, stream)
    torch.cuda.synchronize()
    bank.timing_begin()
    t0 = time.perf_counter()
    for _ in range(steps):
        bank.process_device(io.data_ptr(), N, stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    launches, kms = bank.timing_end()
    res = dict(patch=patch, instances=K, block=N, steps=steps, value=K * N * steps / dt, ms_per_step=1e3 * dt / steps, kernel_ms=kms / launches,
               ring_bytes_per_instance=bank.state_bytes, alg_bytes_per_launch=K * N * FX_BYTES_PER_SAMPLE[patch], finite=bool(torch.isfinite(io).all().item()))
    res["hbm_gbs"] = res["alg_bytes_per_launch"] / (1e-3 * res["kernel_ms"]) / 1e9
    bank.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--cpu-budget", type=float, default=4.0)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    ko = load_oracle()
    rows = []

    def emit(r):
        rows.append(r)
        print(json.dumps(r), flush=True)

    want = (lambda n: args.only is None or n in args.only.split(","))
    if want("cfg1"):
        r = gpu_synth("sine", 1, 1024, 64, 4); r["config"] = "cfg1: 1-voice Fast::Sine, N=1024"; r["cpu_1core"] = cpu_synth(ko, "sine", 1024, args.cpu_budget); emit(r)
    if want("cfg2"):
        r = gpu_synth("sub2a", 1024, 256, 200, 30); r["config"] = "cfg2a: 1024-voice Subtractive (Saw>>LPF>>ADSR), N=256, sustain"; r["cpu_1core"] = cpu_synth(ko, "sub2a", 256, args.cpu_budget); emit(r)
        r = gpu_synth("sub2a", 1 << 20, 256, 100, 30); r["config"] = "cfg2a x1024: 1,048,576 voices, sustain"; emit(r)
        r = gpu_synth("sub2a", 1 << 20, 256, 15, 2); r["config"] = "cfg2a x1024: 1,048,576 voices, attack/decay ramps (steps 2..17)"; emit(r)
        r = gpu_synth("sub2a", 1 << 20, 256, 20, 30, release=True); r["config"] = "cfg2a x1024: 1,048,576 voices, release ramps"; emit(r)
        r = gpu_synth("sub2b", 1024, 256, 100, 10); r["config"] = "cfg2b: 1024-voice shipped subtractive.k (swept cutoff), N=256"; r["cpu_1core"] = cpu_synth(ko, "sub2b", 256, args.cpu_budget); emit(r)
        r = gpu_synth("sub2b", 1 << 18, 256, 30, 10); r["config"] = "cfg2b: 262,144 voices"; emit(r)
    if want("cfg3"):
        r = gpu_synth("supersaw", 16384, 256, 100, 30); r["config"] = "cfg3: 16384-voice SuperSaw.k, N=256"; r["cpu_1core"] = cpu_synth(ko, "supersaw", 256, args.cpu_budget); emit(r)
        r = gpu_synth("supersaw", 1 << 19, 256, 30, 30); r["config"] = "cfg3 x32: 524,288 voices"; emit(r)
    if want("cfg5"):
        r = gpu_synth("fm4", 131072, 256, 50, 30); r["config"] = "cfg5 (per-GPU share): 131,072-voice 4-operator FM, N=256"; r["cpu_1core"] = cpu_synth(ko, "fm4", 256, args.cpu_budget); emit(r)
        r = gpu_synth("fm3", 131072, 256, 50, 30); r["config"] = "FM.k (3 operators): 131,072 voices"; r["cpu_1core"] = cpu_synth(ko, "fm3", 256, args.cpu_budget); emit(r)
    if want("cfg4"):
        for K in ((256, 4096, 65536) if not args.quick else (256,)):
            r = gpu_fx("pingpong", K, 256, 40, 5); r["config"] = f"cfg4: {K} x PingPong.k, N=256"; emit(r)
        rows[-1]["cpu_1core"] = cpu_pp = cpu_fx(ko, "pingpong", 256, args.cpu_budget)
        for K in ((256, 4096, 16384) if not args.quick else (64,)):
            r = gpu_fx("reverb", K, 256, 20, 3); r["config"] = f"cfg4: {K} x Reverb.k, N=256"; emit(r)
        rows[-1]["cpu_1core"] = cpu_fx(ko, "reverb", 256, args.cpu_budget)
        print(json.dumps({"cpu_pingpong_1core": cpu_pp, "cpu_reverb_1core": rows[-1]["cpu_1core"]}))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
