#!/usr/bin/env python3
"""bench.py — voice*samples/s of the Subtractive patch (BASELINE.json metric) on N MI355X of one node.

A "step" is one audio block (256 samples @ 48 kHz) of every voice resident on the GPU: one klg_render launch
(+ the tiny partial-sum reduce) accumulating into a device-resident stereo block; with N > 1 the voices are
sharded across ranks (independent voice ranges, weak scaling) and the only exchange is one RCCL all-reduce of
the [2][256] stereo block per step (BASELINE.json north_star).  Inputs (voice state) are resident in HBM when the
timed region starts.

Prints ONE JSON line (driver contract) with the extra `roofline` and `cpu_baseline` objects.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3        # fp32 vector peak
FLOPS_PER_VOICE_SAMPLE = {"sub2a": 21, "sub2b": 120, "supersaw": 120, "fm4": 95, "fm3": 72, "sine": 12}   # SURVEY.md §8(d) estimates
# algorithmic HBM bytes per voice per block: record read + words written back (klang_amd/csrc/klg_patches.hpp)
STORE_WORDS = {"sub2a": 8, "sub2b": 19, "supersaw": 12, "fm3": 20, "fm4": 25, "sine": 2}


def pmc_traffic(patch, voices):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json: FETCH_SIZE x 2 per the
    gfx950 half-count correction of MI355X_MICROARCH.md + WRITE_SIZE), or None when no matching profile exists."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        e = t.get(f"{patch}:{voices}")
        return e["bytes_per_launch"] if e else None
    except Exception:
        return None


def cpu_baseline(patch, block, budget_s=12.0):
    """The TEST-ONLY oracle (C restatement, bit-exact vs the reference header) timed on ONE host core on a
    bounded sample of the same workload: 128 voices (one Synth instance) x `block` samples x M blocks."""
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True, stdout=subprocess.DEVNULL)
    ko = C.CDLL(os.path.join(ROOT, "oracle", "_build", "libklang_oracle.so"))
    ko.ko_bank_create.restype = C.c_void_p
    ko.ko_bank_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float]
    ko.ko_bank_note_on.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_long]
    ko.ko_bank_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    ko.ko_patch_from_name.argtypes = [C.c_char_p]
    pid = ko.ko_patch_from_name(patch.encode())
    notes = 128 if patch in ("sub2a", "sine", "bsine") else 32
    synths = 128 // notes
    bank = ko.ko_bank_create(pid, synths, notes, C.c_float(48000.0))
    rng = np.random.default_rng(20250314)
    for sy in range(synths):
        for p in rng.integers(36, 97, size=notes):
            ko.ko_bank_note_on(bank, sy, int(p), C.c_float(0.8), 1)
    mix = np.zeros((2, block), np.float32)
    mp = mix.ctypes.data_as(C.c_void_p)
    t0 = time.perf_counter()
    blocks = 0
    while True:
        for _ in range(50):
            ko.ko_bank_process(bank, None, mp, None, block)
        blocks += 50
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": 128 * block * blocks / dt, "unit": "voice*samples/s", "cores": 1, "kind": "port",
            "sample": f"{patch}: 128 voices x {block} samples x {blocks} blocks, oracle/klang_oracle.c -O2 single thread, {os.cpu_count()} host cores present"}


def cpu_reference(patch, block, budget_s=10.0):
    """The GENUINE reference header (oracle/_ref/ref_subtractive: /root/reference/klang.h compiled where it lies, the binary travels)
    on one host core, same bounded workload: 128 voices x `block` samples x M blocks; wall time of the whole run (process start,
    128 note-ons and writing the mixes included: < 1 %).  None when the binary is not there."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_subtractive")
    if patch != "sub2a" or not os.path.exists(exe):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scenario_io import Scenario

    def run(blocks):
        s = Scenario(patch="sub2a", block=block, blocks=blocks, synths=1, notes=128, dump=[])
        rng = np.random.default_rng(20250314)
        for p in rng.integers(36, 97, size=128):
            s.on(0, 0, int(p), 0.8)
        with tempfile.TemporaryDirectory() as d:
            scn, out = os.path.join(d, "s.scn"), os.path.join(d, "o.bin")
            s.save(scn)
            t0 = time.perf_counter()
            subprocess.run([exe, scn, out], check=True, stdout=subprocess.DEVNULL)
            return time.perf_counter() - t0
    probe = run(1000)
    blocks = int(max(1000, min(40000, 1000 * budget_s / max(probe, 1e-3))))
    dt = run(blocks)
    return {"value": 128 * block * blocks / dt, "unit": "voice*samples/s", "cores": 1, "kind": "reference",
            "sample": f"{patch}: 128 voices x {block} samples x {blocks} blocks, the reference's klang.h v0.7.8 (oracle/_ref/ref_subtractive, clang++ -O2) single thread, {os.cpu_count()} host cores present"}


def valu_issue(patch, V, N, kern_s):
    """VALU ISSUE-rate view of the sustain loop of klg_render_sub2a_x2 (the number that actually bounds this kernel): one wave =
    128 voices; per sample its steady-state loop issues 27.75 instructions (4x unrolled: 109 VALU + 2 ds_write2 per 4 samples)
    + 65 per 16-sample mix flush = 31.8 (counted in the ISA); a SIMD issues one wave64 VALU instruction
    per 4 cycles, 1024 SIMDs at the 2.4 GHz peak engine clock."""
    if patch != "sub2a" or os.environ.get("KLG_RENDER_X1") == "1":
        return {}
    achieved = (V / 128.0) * N * 31.8 / kern_s
    peak = 1024 * 2.4e9 / 4.0
    return {"issue_rate_frac_est": achieved / peak, "wave_instr_per_wave_sample": 31.8, "issue_peak_wave_instr_per_s": peak}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--patch", default="sub2a")
    ap.add_argument("--voices", type=int, default=1 << 22, help="voices per GPU (weak scaling); 4 Mi voices = 336 MB of lane records")
    ap.add_argument("--block", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: klang_amd has no CPU path")
    # test hook: KLG_BENCH_ONE_GPU=1 runs every rank on cuda:0 with the gloo backend so the N > 1 code path (sharding,
    # barriers, max-over-ranks timing, all-reduce of the mix) can be smoke-tested on a 1-GPU box.  Never set by the driver.
    one_gpu = os.environ.get("KLG_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import klang_amd
    notes = 128 if args.patch in ("sub2a", "sine", "bsine") else 32
    synths = max(1, args.voices // notes)
    # weak scaling: every rank owns `synths` instances of the global bank (klang_amd/shard.py: contiguous ranges)
    bank = klang_amd.ShardedSynthBank(args.patch, synths * world, notes, fs=48000.0, max_block=args.block, rank=rank, world=world, device=local_rank)
    V, N = bank.voices, args.block

    # synthetic MIDI / random-parameter workload: every voice sounding, uniform pitches 36..96
    rng = np.random.default_rng(20250314 + rank)
    pitches = rng.integers(36, 97, size=V)
    vels = rng.uniform(0.25, 1.0, size=V)
    bank.random(rank + 1)
    owner = bank.lo + np.arange(V) // notes
    bank.note_on_many(owner, pitches, vels)
    # Throughput mode: blocks are pipelined through a small ring of output buffers, so the (2 KiB, latency-bound) all-reduce
    # of block i overlaps the render of block i+1; a buffer is reused only after its own all-reduce has completed.
    RING = 4
    mixes = [torch.zeros((2, N), dtype=torch.float32, device="cuda") for _ in range(RING)]
    pending = [None] * RING
    stream = torch.cuda.current_stream().cuda_stream
    state = {"i": 0}

    def step():
        k = state["i"] % RING
        state["i"] += 1
        if pending[k] is not None:
            pending[k].wait()                        # the current stream waits for that buffer's collective (issued RING steps ago)
            pending[k] = None
        mixes[k].zero_()
        pending[k] = bank.process_device(mixes[k], N, stream, async_reduce=True)   # render + (world > 1) one RCCL all-reduce of the [2][N] block

    def drain():
        for k in range(RING):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    # settle: run the attack + decay segments (0.01 s + 0.105 s = 22 blocks) untimed so the timed region measures
    # the steady state of a sounding voice (sustain); the all-voices-ramping worst case is measured separately below
    for _ in range(24):
        step()
    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    bank.bank.timing_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                                          # every block of the timed region is fully reduced inside the bracket
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    launches, kernel_ms = bank.bank.timing_end()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    checksum = float(mixes[(state["i"] - 1) % RING].abs().sum().item())

    # worst case for the envelope code: every voice in its release ramp (not part of `value`)
    bank.note_off_many(owner, pitches, np.zeros(V, np.float32))
    step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    for _ in range(20):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt_release = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([dt_release], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_release = float(t.item())

    if rank == 0:
        total_voices = V * world
        value = total_voices * N * args.steps / dt
        ms_per_step = 1e3 * dt / args.steps
        kern_s = 1e-3 * kernel_ms / max(1, launches)
        rec_bytes = bank.bank.state_bytes
        alg_bytes = V * (rec_bytes + 4 * STORE_WORDS.get(args.patch, rec_bytes // 4)) + 2 * N * 4
        achieved = alg_bytes / kern_s / 1e9
        flops = FLOPS_PER_VOICE_SAMPLE.get(args.patch, 0) * V * N / kern_s / 1e12
        out = {
            "metric": "voice*samples/s @48kHz Subtractive",
            "value": value, "unit": "voice*samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.patch}: {V} voices/GPU (Saw>>Biquad LPF>>ADSR), {N}-sample blocks @48kHz, all voices sounding, stereo mix resident in HBM",
                       "voices_per_gpu": V, "block": N, "parallelism": f"voice-shard x{world}" + (" + RCCL all-reduce of [2][%d] per block" % N if world > 1 else ""),
                       "phase": "sustain (all voices held)", "value_all_voices_in_release_ramp": total_voices * N * 20 / dt_release,
                       "realtime_voices_equiv": int(value / 48000.0), "realtime_voices_equiv_worst_case": int(total_voices * N * 20 / dt_release / 48000.0),
                       "block_deadline_ms": 1e3 * N / 48000.0, "mix_checksum": checksum},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args.patch, V),
                         "kernel": ("klg_render_sub2a_x2<false>" if args.patch == "sub2a" and os.environ.get("KLG_RENDER_X1") != "1" else "klg_render<%s>" % args.patch), "kernel_ms": 1e3 * kern_s, "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "synth patches keep voice state in registers: the binding unit is fp32 VALU issue, see `valu`",
                         "valu": {"achieved_tflops_est": flops, "peak_tflops": FP32_PEAK_TFLOPS, "frac": flops / FP32_PEAK_TFLOPS,
                                  "flops_per_voice_sample_est": FLOPS_PER_VOICE_SAMPLE.get(args.patch, 0), **valu_issue(args.patch, V, N, kern_s)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            port = cpu_baseline(args.patch, N, budget_s=8.0)                 # the C restatement (oracle/klang_oracle.c)
            try:
                ref = cpu_reference(args.patch, N, budget_s=8.0)             # the genuine header, where its binary travelled
            except Exception as e:                                          # a baseline must never cost the bench line
                print(f"bench.py: reference baseline unavailable ({e}); reporting the port", file=sys.stderr)
                ref = None
            out["cpu_baseline"] = dict(ref, port_value=port["value"], port_sample=port["sample"]) if ref else port
        print(json.dumps(out))
    bank.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
