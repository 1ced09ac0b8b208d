#!/usr/bin/env python3
"""bench.py — voice*samples/s of the Subtractive patch (BASELINE.json metric) on N MI355X of one node, and every BASELINE config beside it.

HEADLINE (`value`): SURVEY.md §8(d)'s config-2 event script — Saw >> LPF >> ADSR voices, 256-sample blocks @ 48 kHz, a 375-block (2 s) note
life: note-on at block 0, note-off at block 150 + (v mod 64), the 0.255 s release, silence — played as a STEADY STATE: the bank is 375 voice
groups, group g running the script g blocks late and starting over when it ends, so every block holds every phase of the script in the
script's own proportions (inside a group all voices run in phase, so a wave sees exactly what it sees in the literal script: all-sustain,
then 64 staggered releases, then silence).  A "step" is one block: the block's note events are applied from an event script resident in HBM
(klg_script_*: every on() ran on the host before the timed region), then ONE klg_render launch + the partial-sum reduce accumulate the
stereo block in HBM.  value = sounding voices x samples / time (SURVEY §8d "active voices").  With N > 1 every rank owns its own shard of
voices (weak scaling) and the only exchange is one RCCL all-reduce of the [2][256] block per step.

`configs` (N = 1 only): each BASELINE config at its own size, self-timed in the same run — cfg 2 at 1024 voices, cfg 3 (16384 SuperSaw
voices), cfg 4 (4096 PingPong / 4096 Reverb instances, each with its own roofline), the cfg-5 per-GPU share (131072 FM4 voices) — the
literal script at the headline size with its phases (all-ramping, sustain-only, staggered release), and the p99 real-time deadline test.

Prints ONE JSON line (driver contract) with `roofline`, `cpu_baseline` and `configs`.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3        # fp32 vector peak (packed FMA)
FLOPS_PER_VOICE_SAMPLE = {"sub2a": 21, "sub2b": 120, "supersaw": 120, "fm4": 95, "fm3": 72, "sine": 12}   # SURVEY.md §8(d) estimates
# algorithmic HBM bytes per voice per block: record read + words written back (klang_amd/csrc/klg_patches.hpp)
STORE_WORDS = {"sub2a": 8, "sub2b": 19, "supersaw": 12, "fm3": 20, "fm4": 25, "sine": 2}
FX_BYTES_PER_SAMPLE = {"pingpong": 32, "reverb": 312}     # SURVEY.md §8(d) algorithmic bytes per instance*sample
PATCH_ID = {"sine": 0, "bsine": 1, "sub2a": 2, "sub2b": 3, "supersaw": 4, "fm3": 5, "fm4": 6}          # include/klang_mi355.h klg_patch
NOTES = {"sub2a": 128, "sine": 128, "bsine": 128}          # note slots per Synth instance (others: 32)
SCRIPT_BLOCKS, OFF_BLOCK, OFF_SPREAD = 375, 150, 64        # SURVEY §8(d) cfg 2: 2 s, note-off at 150 + (v mod 64)
# blocks a voice still sounds after its note-off: ceil(release time * 48000 / 256) — sub2a 0.25 s + 5 ms = 12240 samples, SuperSaw.k 0.5 s + 5 ms, FM 1 s + 5 ms
RELEASE_BLOCKS = {"sub2a": 48, "supersaw": 95, "fm3": 189, "fm4": 189}
KERNEL_OF = {"sub2a": "klg_render_sub2a_x2", "supersaw": "klg_render_supersaw_sp", "fm4": "klg_render<klg::PatchFM<4>", "fm3": "klg_render<klg::PatchFM<3>",
             "pingpong": "klg_fx_pingpong_x", "reverb": "klg_fx_reverb_q"}


# ---------------------------------------------------------------------------------------------------------------------------------
# the event script of SURVEY §8(d), as an HBM-resident klg_script
# ---------------------------------------------------------------------------------------------------------------------------------
def build_script(bank, groups, rng, cyclic):
    """Voices [0, V) in `groups` equal contiguous groups; group g starts its note at block g * (375 // groups) ... (cyclic) or every
    group at block 0 (literal script, groups == 1).  Returns (EventScript, sounding[b] = voices sounding during block b once settled);
    sounding.alive_after[b] = voices still sounding AFTER block b (a voice's last block is the one in which its envelope runs out)."""
    import klang_amd
    V, notes = bank.voices, bank.notes
    pitches = rng.integers(36, 97, size=V).astype(np.int32)
    vels = rng.uniform(0.25, 1.0, size=V).astype(np.float32)
    owner = (np.arange(V) // notes).astype(np.int32)
    records = bank.note_records(owner, pitches, vels)                       # every on() runs HERE, once (host)
    script = klang_amd.EventScript(bank, SCRIPT_BLOCKS)
    first = script.add_records(records)
    v = np.arange(V, dtype=np.int64)
    gs = V // groups
    g = v // gs
    start = (g * (SCRIPT_BLOCKS // groups)) % SCRIPT_BLOCKS if cyclic else np.zeros(V, np.int64)
    off = start + OFF_BLOCK + (v % OFF_SPREAD)
    if cyclic:
        off %= SCRIPT_BLOCKS
    script.note_on(start, v, first + v)
    script.note_off(off, v)
    script.commit()
    # sounding voices per block: from note-on until RELEASE_BLOCKS blocks after the note-off (the block in which the envelope runs out is rendered)
    life = OFF_BLOCK + (v % OFF_SPREAD) + RELEASE_BLOCKS[bank.patch]                     # blocks a voice sounds, counted from its note-on
    class Sounding(np.ndarray):
        pass
    sounding = np.zeros(SCRIPT_BLOCKS, np.int64).view(Sounding)
    sounding.alive_after = np.zeros(SCRIPT_BLOCKS, np.int64)
    for b in range(SCRIPT_BLOCKS):
        local = (b - start) % SCRIPT_BLOCKS if cyclic else b - start
        sounding[b] = int(((local >= 0) & (local < life)).sum())
        sounding.alive_after[b] = int(((local >= 0) & (local < life - 1)).sum())
    return script, sounding


def alg_bytes(patch, bank, live_voices, n):
    return live_voices * (bank.state_bytes + 4 * STORE_WORDS.get(patch, bank.state_bytes // 4)) + 2 * n * 4


# ---------------------------------------------------------------------------------------------------------------------------------
# CPU baselines (reported beside the GPU number, N = 1 only)
# ---------------------------------------------------------------------------------------------------------------------------------
def cpu_baseline(patch, block, budget_s=12.0):
    """The TEST-ONLY oracle (C restatement, bit-exact vs the reference header) timed on ONE host core on a
    bounded sample of the same workload: 128 voices (one Synth instance) x `block` samples x M blocks."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True, stdout=subprocess.DEVNULL)
    ko = C.CDLL(os.path.join(ROOT, "oracle", "_build", "libklang_oracle.so"))
    ko.ko_bank_create.restype = C.c_void_p
    ko.ko_bank_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float]
    ko.ko_bank_note_on.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_long]
    ko.ko_bank_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    ko.ko_patch_from_name.argtypes = [C.c_char_p]
    pid = ko.ko_patch_from_name(patch.encode())
    notes = NOTES.get(patch, 32)
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
            "sample": f"{patch}: 128 voices x {block} samples x {blocks} blocks, all sounding, oracle/klang_oracle.c -O2 single thread, {os.cpu_count()} host cores present"}


def cpu_reference(patch, block, budget_s=10.0):
    """The GENUINE reference header (oracle/_ref/ref_subtractive: /root/reference/klang.h compiled where it lies, the binary travels)
    on one host core, same bounded workload: 128 voices x `block` samples x M blocks; wall time of the whole run (process start,
    128 note-ons and writing the mixes included: < 1 %).  None when the binary is not there."""
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
    return {"value": 128 * block * blocks / dt, "unit": "voice*samples/s", "cores": 1, "kind": "reference", "blocks": blocks,
            "sample": f"{patch}: 128 voices x {block} samples x {blocks} blocks, all sounding, the reference's klang.h v0.7.8 (oracle/_ref/ref_subtractive, clang++ -O2) single thread, {os.cpu_count()} host cores present"}


def host_cores():
    """(cpu ids to pin to: one per PHYSICAL core this process may use, note).  The GPU box's cgroup grants fewer CPUs than the host has
    (DESIGN.md §7): the affinity mask and cpu.max are what count, not os.cpu_count()."""
    allowed = sorted(os.sched_getaffinity(0))
    seen, picked = set(), []
    for c in allowed:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib); picked.append(c)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(float(q) / float(per)))
    except (OSError, ValueError):
        pass
    if quota is not None and quota < len(picked):
        picked = picked[:quota]
    return picked, f"{os.cpu_count()} logical CPUs present, {len(allowed)} in this process's affinity mask, cgroup cpu.max {'= ' + str(quota) if quota else 'unlimited'}, {len(picked)} physical cores used"


def cpu_reference_node(patch, block, blocks):
    """SURVEY §8(d)(b) / north_star "timed on the node's own host cores (core count stated)": ONE pinned process of the genuine-header binary
    per physical core this job may use, all started together, each rendering the single-core sample (128 sounding voices x `blocks` blocks);
    value = all voice*samples / wall time of the slowest.  None when the binary is not there."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_subtractive")
    if patch != "sub2a" or not os.path.exists(exe):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scenario_io import Scenario
    cores, note = host_cores()
    with tempfile.TemporaryDirectory() as d:
        procs = []
        for i, c in enumerate(cores):
            s = Scenario(patch="sub2a", block=block, blocks=blocks, synths=1, notes=128, dump=[])
            for pch in np.random.default_rng(20250314 + i).integers(36, 97, size=128):
                s.on(0, 0, int(pch), 0.8)
            s.save(os.path.join(d, f"s{i}.scn"))
        t0 = time.perf_counter()
        for i, c in enumerate(cores):
            procs.append(subprocess.Popen([exe, os.path.join(d, f"s{i}.scn"), os.path.join(d, f"o{i}.bin")], stdout=subprocess.DEVNULL,
                                          preexec_fn=(lambda c=c: os.sched_setaffinity(0, {c}))))
        rc = [p.wait() for p in procs]
        dt = time.perf_counter() - t0
    if any(rc):
        return None
    total = len(cores) * 128 * block * blocks / dt
    return {"value": total, "unit": "voice*samples/s", "cores": len(cores), "per_core": total / len(cores), "kind": "reference",
            "sample": f"{patch}: one process per physical core, each 128 voices x {block} samples x {blocks} blocks, all sounding, the reference's klang.h v0.7.8 (oracle/_ref/ref_subtractive), pinned; {note}"}


# VALU wave-instructions per voice*sample of the two other synth kernels, from their counters (SQ_INSTS_VALU per launch / voices x samples;
# they do not depend on the data: the sustain loop's paths are wave-uniform)
VALU_INSTR_PER_VOICE_SAMPLE = {"fm4": (48537600 / (131072 * 256), "profiles/r04_pmc/pmc_fm4.json: SQ_INSTS_VALU 48,537,600 per launch of 131,072 voices x 256 samples")}


def valu_issue(wave_samples, kern_s, per_sample):
    """VALU ISSUE-rate view of klg_render_sub2a_x2 (the unit that binds it): one wave = 128 voices; `per_sample` wave-instructions per
    wave*sample (counted in the ISA); a SIMD issues one wave64 VALU instruction per 4 cycles, 1024 SIMDs at 2.4 GHz."""
    peak = 1024 * 2.4e9 / 4.0
    return {"issue_rate_frac_est": wave_samples * per_sample / kern_s / peak, "wave_instr_per_wave_sample": per_sample, "issue_peak_wave_instr_per_s": peak}


# ---------------------------------------------------------------------------------------------------------------------------------
# live HBM traffic of the dominant kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) around a child of this script
# ---------------------------------------------------------------------------------------------------------------------------------
def pmc_traffic_live(args, kernel_substr, timeout_s=240, child_args=None, blocks_per_launch=1, total_blocks=None):
    """bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KB * 1024 (gfx950 FETCH_SIZE half-count, MI355X_MICROARCH.md §HBM), mean over the
    child's steady-state launches of the named kernel; None if rocprofv3 is unavailable or anything goes wrong (never costs the bench line)."""
    import csv
    import glob
    import shutil
    if os.environ.get("KLG_BENCH_PMC", "1") == "0" or not shutil.which("rocprofv3"):
        return None, "rocprofv3 not run"
    got = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="klg_pmc_", dir="/tmp")
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child"] + (child_args if child_args is not None else ["--voices", str(args.voices), "--block", str(args.block), "--patch", args.patch])
            subprocess.run(cmd, check=True, timeout=timeout_s, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
                        vals.append(float(r["Counter_Value"]))
            shutil.rmtree(d, ignore_errors=True)
            if not vals:
                return None, f"no {counter} rows for {kernel_substr}"
            if total_blocks:                                            # a span may be several launches of the kernel (PingPong: two): every launch of the child over every block of the child
                got[counter] = sum(vals) / total_blocks * blocks_per_launch
            else:
                vals = vals[len(vals) // 2:]                            # the child's settled blocks
                got[counter] = sum(vals) / len(vals)
        return (2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"]) * 1024.0 / blocks_per_launch, f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in this run (2 passes, {len(vals)} launches each" + (f" of {blocks_per_launch} blocks" if blocks_per_launch > 1 else "") + "); (2*FETCH_SIZE + WRITE_SIZE) KB * 1024"
    except Exception as e:                                              # noqa: BLE001
        return None, f"pmc leg failed: {type(e).__name__}"


PMC_FX_SPAN = 16                 # blocks per span in the effect legs' counter children
PMC_FX_SPANS = 12                # ... and spans per child


def random_dials(bank, K):
    """one PingPong instance in seven off its defaults: Feedback, Delay (down to the 1 ms at which the right tap is 24 samples behind the cursor), Scratch and
    Rate (vibrato), the Delay target — what tools/pingpong_recorded_bench.py sets"""
    rng = np.random.default_rng(3)
    for k in range(0, K, 7):
        for c, lo, hi in ((0, 0.2, 0.9), (1, 0.01, 0.6), (2, 0.0, 1.0), (3, 0.01, 1.0), (5, 0.0, 0.4)):
            bank.set_control(k, c, float(rng.uniform(lo, hi)))


def run_fx_recorded(K, N, dials=None, tag=""):
    """f1, the GENERATED path at config 4's size: the unchanged examples/PingPong.k as the facade records it (tests/golden/pingpong_recorded.klgg + the record a
    fresh object packs to) — the hipRTC-compiled sample-parallel kernel of klg_graph_staged.hpp —, one block per call like a real-time host, against the
    hand-written kernel's bytes.  200 blocks after 60 untimed ones (a fresh object's dial smoothers have arrived); random dials: see random_dials()."""
    import torch
    import klang_amd
    root = os.path.dirname(os.path.abspath(__file__))
    prog = open(os.path.join(root, "tests", "golden", "pingpong_recorded.klgg")).read()
    rec = np.array([int(w, 16) for w in open(os.path.join(root, "tests", "golden", "pingpong_recorded.rec")).read().split()], np.uint32)
    bank = klang_amd.FxBank(prog, K, max_block=N, initial_record=rec, channels=2)
    form = bank.graph_form()
    if dials == "random7":
        random_dials(bank, K)
    g = torch.Generator(device="cuda").manual_seed(1)
    io = torch.rand((K, 2, N), device="cuda", generator=g) - 0.5
    ts = torch.cuda.Stream()
    blocks = 200
    with torch.cuda.stream(ts):
        st = ts.cuda_stream
        for _ in range(60):
            bank.process_device(io.data_ptr(), N, st)
        torch.cuda.synchronize()
        bank.timing_begin()
        t0 = time.perf_counter()
        for _ in range(blocks):
            bank.process_device(io.data_ptr(), N, st)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    launches, ms = bank.timing_end()
    kern_s = 1e-3 * ms / blocks
    ab = K * N * FX_BYTES_PER_SAMPLE["pingpong"]
    res = {"name": f"cfg4_pingpong_{K}_recorded{tag}", "workload": ("one instance in seven with random dials: " if dials == "random7" else "") + f"{K} x the unchanged examples/PingPong.k RECORDED (generated kernel klg_fx_staged: {form}), {blocks} blocks of {N} samples, one per call",
           "value": K * N * blocks / dt, "unit": "instance*samples/s", "ms_per_step": 1e3 * dt / blocks, "steps": blocks, "kernel_ms_mean": 1e3 * kern_s, "finite": bool(torch.isfinite(io).all().item()),
           "roofline": {"bound": "hbm", "achieved": ab / kern_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ab / kern_s / 1e9 / HBM_PEAK_GBS, "traffic": None, "traffic_source": "not collected",
                        "kernel": "klg_fx_staged (hipRTC, generated from the recorded program)", "algorithmic_bytes_per_launch": ab, "bytes_per_instance_sample": FX_BYTES_PER_SAMPLE["pingpong"], "per": f"block of {N} samples, one launch"}}
    bank.close()
    return res


def pmc_child_fx(spec, N):
    """what rocprofv3 wraps for an effect leg: `patch:K[:ctl=value,...]` — the bank with its dials at rest (the spans before the counted ones let a
    PingPong's smoothers converge), spans of PMC_FX_SPAN blocks through klg_fx_render_device"""
    import torch
    import klang_amd
    parts = spec.split(":")
    patch, K = parts[0], int(parts[1])
    bank = klang_amd.FxBank(patch, K, max_block=N)
    if len(parts) > 2 and parts[2] == "random7":
        random_dials(bank, K)
    else:
        for kv in (parts[2].split(",") if len(parts) > 2 and parts[2] else []):
            c, v = kv.split("=")
            for k in range(K):
                bank.set_control(k, int(c), float(v))
    io = (torch.rand((PMC_FX_SPAN, K, 2, N), device="cuda") - 0.5) * 0.1
    ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
    for _ in range(PMC_FX_SPANS):                                          # (PingPong: every launch of every span is counted — a span is one or two launches —; Reverb: the later half)
        bank.render_device(io.data_ptr(), PMC_FX_SPAN, N, ts.cuda_stream)
    torch.cuda.synchronize()
    bank.close()


def pmc_child(args):
    """what rocprofv3 wraps: the headline steady state, few blocks (setup identical, so the kernel sees the same voice population)"""
    import torch
    import klang_amd
    if args.pmc_fx:
        return pmc_child_fx(args.pmc_fx, args.block)
    notes = NOTES.get(args.patch, 32)
    V = groups_voices(args.voices)
    bank = klang_amd.SynthBank(args.patch, synths=V // notes, notes=notes, max_block=args.block)
    script, _ = build_script(bank, SCRIPT_BLOCKS, np.random.default_rng(20250314), cyclic=True)
    mix = torch.zeros((2, args.block), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
    st = ts.cuda_stream
    for b in range(SCRIPT_BLOCKS + 24):
        mix.zero_(); script.play_device(b % SCRIPT_BLOCKS, mix.data_ptr(), args.block, st)
    torch.cuda.synchronize()
    bank.close()


def groups_voices(voices):
    """375 equal groups of whole workgroups (512 voices of the packed kernel)"""
    gs = max(512, (voices // SCRIPT_BLOCKS) // 512 * 512)
    return gs * SCRIPT_BLOCKS


# ---------------------------------------------------------------------------------------------------------------------------------
# the other BASELINE configs, each at its own size (N = 1)
# ---------------------------------------------------------------------------------------------------------------------------------
def run_literal_script(patch, voices, N, label, phases=False):
    """the §8(d) script as written (every voice starts at block 0), 375 blocks through klg_script_play_device; value = sounding voices x samples / time"""
    import torch
    import klang_amd
    notes = NOTES.get(patch, 32)
    bank = klang_amd.SynthBank(patch, synths=max(1, voices // notes), notes=notes, max_block=N)
    bank.random(12345)
    script, sounding = build_script(bank, 1, np.random.default_rng(20250314), cyclic=False)
    mix = torch.zeros((2, N), dtype=torch.float32, device="cuda")
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(SCRIPT_BLOCKS)]
    torch.cuda.synchronize()
    ts = torch.cuda.Stream()                                               # the library launches on THIS stream (a non-default handle), and so do the events
    with torch.cuda.stream(ts):
        st = ts.cuda_stream
        mix.zero_(); torch.cuda.synchronize()
        bank.timing_begin()
        t0 = time.perf_counter()
        for b in range(SCRIPT_BLOCKS):
            ev[b][0].record()
            mix.zero_(); script.play_device(b, mix.data_ptr(), N, st)
            ev[b][1].record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    launches, kms = bank.timing_end()
    ms = np.array([a.elapsed_time(b) for a, b in ev])
    alive = int((bank.stages() != 3).sum())
    V = bank.voices
    # the same 375 blocks as ONE library call (klg_script_render_device: the span cleared once, then one launch per block — for banks of a few
    # workgroups the events and the voice sum are inside the render launch): what the blocks cost without a Python call between them
    one_call = None
    if voices <= 262144:
        out = torch.zeros((SCRIPT_BLOCKS, 2, N), dtype=torch.float32, device="cuda")
        with torch.cuda.stream(ts):
            script.render_device(0, SCRIPT_BLOCKS, out.data_ptr(), N, ts.cuda_stream)      # (warm: the voices play the script again from its block 0; the span's launches are captured as a hipGraph)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            script.render_device(0, SCRIPT_BLOCKS, out.data_ptr(), N, ts.cuda_stream)      # the timed pass: the graph replayed
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
            bank.timing_begin()                                                              # (kernel time: a third pass, launched one by one — event pairs cannot sit inside a replayed graph)
            script.render_device(0, SCRIPT_BLOCKS, out.data_ptr(), N, ts.cuda_stream)
            torch.cuda.synchronize()
        l1, kms1 = bank.timing_end()
        one_call = {"ms_per_step": 1e3 * dt1 / SCRIPT_BLOCKS, "kernel_ms_mean": kms1 / max(1, l1), "value": float(sounding.sum()) * N / dt1,
                    "what": "klg_script_render_device(0, 375), its launches captured once as a hipGraph and replayed: one clear of the [375][2][N] span, then per block ONE launch for banks of a few workgroups (events + render + voice sum inside it; larger banks: events / render / reduce), no host call between blocks",
                    "mix_abs_sum": float(out.abs().sum().item())}
    res = {"name": label, "workload": f"{patch}: {V} voices, SURVEY 8(d) script as written: note-on at block 0, note-off at block 150 + (v mod 64), {SCRIPT_BLOCKS} blocks of {N} samples, events from HBM",
           "value": float(sounding.sum()) * N / dt, "unit": "voice*samples/s (sounding voices)", "ms_per_step": 1e3 * dt / SCRIPT_BLOCKS, "steps": SCRIPT_BLOCKS,
           "kernel_ms_mean": kms / max(1, launches), "voices_sounding_mean": float(sounding.mean()), "voices_alive_at_end": alive,
           "ms_per_block_sustain": float(np.median(ms[40:OFF_BLOCK])), "value_sustain_phase": V * N / (1e-3 * float(np.median(ms[40:OFF_BLOCK]))),
           # every block of the script — attack, sustain, the staggered releases — between its own pair of events on the stream the library launches on
           "ms_per_block_max": float(ms.max()), "ms_per_block_max_at": int(ms.argmax()), "block_deadline_ms": 1e3 * N / 48000.0, "every_block_within_the_deadline": bool(ms.max() <= 1e3 * N / 48000.0)}
    if one_call:
        res["python_loop"] = {"ms_per_step": res["ms_per_step"], "value": res["value"], "kernel_ms_mean": res["kernel_ms_mean"], "what": "one Python -> C-ABI call per block (mix.zero_() + klg_script_play_device): host submission bound at this size"}
        res.update(ms_per_step=one_call["ms_per_step"], value=one_call["value"], kernel_ms_mean=one_call["kernel_ms_mean"], one_call=one_call)
    kern_s = 1e-3 * float(np.median(ms[40:OFF_BLOCK]))
    ab = alg_bytes(patch, bank, V, N)
    tf = FLOPS_PER_VOICE_SAMPLE.get(patch, 0) * V * N / kern_s / 1e12
    res["roofline"] = {"bound": "valu", "achieved": tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_PEAK_TFLOPS, "traffic": None,
                       "flops_per_voice_sample_est": FLOPS_PER_VOICE_SAMPLE.get(patch, 0),
                       "kernel": KERNEL_OF.get(patch, patch), "phase": "sustain (block time incl. event kernel + reduce)",
                       "hbm": {"bound": "hbm", "achieved": ab / kern_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ab / kern_s / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": ab},
                       "note": "voice state lives in registers: VALU-issue bound by design (DESIGN.md §3); 157.3 TFLOP/s is the packed-FMA peak, separate mul / add (bit-parity) reach at most half"}
    if patch in VALU_INSTR_PER_VOICE_SAMPLE and V == {"fm4": 131072, "supersaw": 16384}[patch]:
        per, src = VALU_INSTR_PER_VOICE_SAMPLE[patch]
        peak = 1024 * 2.4e9 / 4.0
        kms_total = kms1 if one_call else kms                                   # the render kernels of all 375 blocks (HIP events on their dispatches)
        res["roofline"]["valu"] = {"issue_rate_frac_est": float(sounding.sum()) * N * per / (1e-3 * kms_total) / peak, "over": "the script's 375 render kernels: sounding voice*samples x instructions per voice*sample / their summed duration",
                                   "wave_instr_per_voice_sample": per, "issue_peak_wave_instr_per_s": peak, "counted": src,
                                   "reading": "the fraction of the chip's VALU issue slots (one wave64 instruction per SIMD and 4 cycles) the render kernel's own instructions fill; SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES of the same file is that figure PER WAVE (several waves share a SIMD's slots)"}
    if phases:
        res["phases_ms_per_block"] = {"attack_decay_all_ramping(1..21)": float(np.median(ms[1:22])), "sustain_only(40..149)": float(np.median(ms[40:OFF_BLOCK])),
                                      "staggered_release(150..213)": float(np.median(ms[OFF_BLOCK:OFF_BLOCK + OFF_SPREAD])), "release_tail(214..260)": float(np.median(ms[214:261])),
                                      "all_off(270..374)": float(np.median(ms[270:]))}
        res["value_all_ramping_phase"] = V * N / (1e-3 * float(np.median(ms[1:22])))
    script.close(); bank.close()
    return res


def run_fx(patch, K, N, dials=None, tag="", args=None, per_block=False):
    """cfg 4: K instances, 375 blocks (2 s): a white-noise burst for the first 4800 samples, then silence; io resident in HBM.
    The blocks are submitted as SPANS (klg_fx_render_device: the host knows all its blocks — the effect template's callback,
    templates/juce/effect/Source/PluginProcessor.cpp:153-178, called span after span): the span's [blocks][K][2][N] buffer is filled (burst or
    silence) inside the timed region, then processed in place.  dials: {index: value} set on every instance before the first block."""
    import torch
    import klang_amd
    bank = klang_amd.FxBank(patch, K, max_block=N)
    if dials == "random7":
        random_dials(bank, K)
    else:
        for c, v in (dials or {}).items():
            for k in range(K):
                bank.set_control(k, c, v)
    g = torch.Generator(device="cuda").manual_seed(1)
    burst_blocks = (4800 + N - 1) // N
    inputs = torch.rand((burst_blocks, K, 2, N), device="cuda", generator=g) - 0.5
    inputs[-1, :, :, 4800 - (burst_blocks - 1) * N:] = 0
    block_bytes = K * 2 * N * 4
    span = next(d for d in (75, 25, 15, 5, 3, 1) if d * block_bytes <= (2 << 30) or d == 1)   # the script's first fifth (75 blocks): spans of a third of this (below)
    # the other four fifths: as few spans as 4 GB of io allow (a host that renders offline hands over what it has; 4,096 instances: all 300 blocks in one call)
    rest_span = next(d for d in (300, 150, 75, 25, 15, 5, 3, 1) if d * block_bytes <= (4 << 30) or d == 1)
    if per_block:                                    # one block per call, as a real-time host hands them over (klg_fx_process_device: nothing runs across a block boundary)
        span = rest_span = 1
    io = torch.zeros((rest_span, K, 2, N), device="cuda")
    torch.cuda.synchronize()
    ts = torch.cuda.Stream()
    with torch.cuda.stream(ts):
        st = ts.cuda_stream
        bank.render_device(io.data_ptr(), min(span, 8), N, st)   # untimed: silence in, silence out — the bank's 6-51 GB of zeroed delay lines are really there afterwards
        torch.cuda.synchronize()
        bank.timing_begin()
        t0 = time.perf_counter()
        head = SCRIPT_BLOCKS // 5                    # the kernel's time is collected in two parts: the first fifth of the script (a PingPong's dial smoothers are
        b0 = 0                                       # still on their way for ~40 blocks after construction: its general pipeline), and the rest (dials at rest).
        while b0 < SCRIPT_BLOCKS:                    # Whether a PingPong launch may run its request-ahead pipeline is decided per launch, on the dials as the launch finds
            if b0 == head:                           # them: the first fifth goes in spans of a third (a launch that starts while a smoother still moves stays general)
                head_launches, head_ms = bank.timing_end()      # (reads the events of finished launches: synchronises the stream once, inside the timed region — counted in dt)
                bank.timing_begin()
            take = min(rest_span if b0 >= head else max(1, span // 3), SCRIPT_BLOCKS - b0, (head - b0) if b0 < head else SCRIPT_BLOCKS)
            io[:take].zero_()                                               # the host's next input: silence ...
            if b0 < burst_blocks:
                nb = min(take, burst_blocks - b0); io[:nb].copy_(inputs[b0:b0 + nb])   # ... or the burst
            if per_block:
                bank.process_device(io.data_ptr(), N, st)
            else:
                bank.render_device(io.data_ptr(), take, N, st)
            b0 += take
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    rest_launches, rest_ms = bank.timing_end()
    kern_s = 1e-3 * (head_ms + rest_ms) / SCRIPT_BLOCKS                     # per BLOCK (a PingPong span is one launch, a Reverb block one or two)
    rest_s = 1e-3 * rest_ms / (SCRIPT_BLOCKS - head)
    ab = K * N * FX_BYTES_PER_SAMPLE[patch]
    kernel = KERNEL_OF[patch]
    traffic, how = None, "not collected"
    spec = f"{patch}:{K}:" + ("random7" if dials == "random7" else ",".join(f"{c}={v}" for c, v in (dials or {}).items()))
    one_launch = patch == "pingpong"
    if args is not None and os.environ.get("KLG_BENCH_PMC_FX", "1") != "0" and not per_block:
        traffic, how = pmc_traffic_live(args, kernel, timeout_s=180, child_args=["--pmc-fx", spec, "--block", str(N)], blocks_per_launch=PMC_FX_SPAN if one_launch else 1,
                                        total_blocks=PMC_FX_SPANS * PMC_FX_SPAN if one_launch else None)
    roof = {"bound": "hbm", "achieved": ab / kern_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ab / kern_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": how,
            "kernel": kernel, "algorithmic_bytes_per_launch": ab, "bytes_per_instance_sample": FX_BYTES_PER_SAMPLE[patch], "per": "block of %d samples (a PingPong span of %d blocks is ONE launch: its duration / %d)" % (N, rest_span, rest_span),
            "after_the_first_fifth": {"kernel_ms_mean": 1e3 * rest_s, "frac": ab / rest_s / 1e9 / HBM_PEAK_GBS, "launches": rest_launches,
                                      "note": "the same blocks without the script's first 75, in which a freshly constructed PingPong's dial smoothers are still converging"}}
    if patch == "reverb":
        # what the algorithm really moves: SURVEY's 312 B count every FilteredDelay once per sample; Reverb.k processes each TWICE (Reverb.k:150-168):
        # reads 20 taps x 2 lines x 4 B + 16 lines x 2 heads x 4 B + io 8 B = 296 B, writes 2 x 4 B + 16 lines x 2 inputs x 4 B + io 8 B = 144 B
        db = K * N * 440
        roof["distinct_bytes_per_launch"] = db
        roof["distinct_bytes_per_instance_sample"] = 440
        roof["frac_on_distinct_bytes"] = db / kern_s / 1e9 / HBM_PEAK_GBS
    res = {"name": f"cfg4_{patch}_{K}{tag}", "workload": (("one instance in seven with random dials: " if dials == "random7" else f"dials {dials}: ") if dials else "") + f"{K} x {patch.capitalize()}.k (Stereo::Effect), {SCRIPT_BLOCKS} blocks of {N} samples in spans of {span} (klg_fx_render_device): noise burst 4800 samples then silence, io + {bank.state_bytes * K / 1e9:.1f} GB of delay lines resident in HBM",
           "value": K * N * SCRIPT_BLOCKS / dt, "unit": "instance*samples/s", "ms_per_step": 1e3 * dt / SCRIPT_BLOCKS, "steps": SCRIPT_BLOCKS, "kernel_ms_mean": 1e3 * kern_s, "blocks_per_span": rest_span, "blocks_per_span_in_the_first_fifth": max(1, span // 3),
           "finite": bool(torch.isfinite(io).all().item()), "roofline": roof}
    bank.close()
    return res

NOISE_NOTE_PROGRAM = """klgg 1
ctl 0
node 0 lpf
op noise 0 -1 -1 -1 1
op lpf 1 0 -1 0 0
op noise 2 -1 -1 -1 0
op const 3 -1 -1 -1 3dcccccd
op mul 4 2 3 -1 0
op add 5 1 4 -1 0
ret 5
end
"""


def run_noise_notes(voices, N, blocks=200):
    """Notes with Noise generators (klang.h:4947-4951, 5357-5366): `hiss >> lpf` + `grit * 0.1` — a Fast::Noise through a biquad and a Basic::Noise,
    two rand() draws per voice and sample, all voices sounding.  The draws come from the C library's sequence continued ON THE DEVICE (klg_rand_fill), in the order
    Synth::process walks the notes.  ms per block as a real-time host sees it (a call and a wait per block) and queued back to back."""
    import torch
    import klang_amd
    P = 128
    bank = klang_amd.SynthBank(NOISE_NOTE_PROGRAM, synths=max(1, voices // P), notes=min(P, voices), max_block=N)
    V, W = bank.voices, bank.state_bytes // 4
    words = np.zeros((V, W), np.uint32)
    words[:, 0] = 1
    words[:, 1:6] = np.array([0.02, 0.04, 0.02, -1.56, 0.64], np.float32).view(np.uint32)
    bank.voices_upload(np.arange(V, dtype=np.int32), words)
    bank.random(1)
    mix = torch.zeros((2, N), dtype=torch.float32, device="cuda")
    ts = torch.cuda.Stream()
    with torch.cuda.stream(ts):
        st = ts.cuda_stream
        for _ in range(5):
            mix.zero_(); bank.process_device(mix.data_ptr(), N, st)
        torch.cuda.synchronize()
        w = np.empty(blocks)
        for b in range(blocks):
            t0 = time.perf_counter()
            mix.zero_(); bank.process_device(mix.data_ptr(), N, st); torch.cuda.synchronize()
            w[b] = 1e3 * (time.perf_counter() - t0)
        t0 = time.perf_counter()
        for _ in range(blocks):
            mix.zero_(); bank.process_device(mix.data_ptr(), N, st)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        bank.timing_begin()
        for _ in range(blocks):
            mix.zero_(); bank.process_device(mix.data_ptr(), N, st)
        torch.cuda.synchronize()
        aux_l, aux_ms = bank.timing_end_aux()
        l, kms = bank.timing_end()
    res = {"name": f"noise_note_{V}", "workload": f"{V} notes of Fast::Noise >> LPF + Basic::Noise (2 rand() per voice and sample, drawn on the device in Synth::process order), {blocks} blocks of {N} samples",
           "value": V * N * blocks / dt, "unit": "voice*samples/s", "ms_per_step": 1e3 * dt / blocks, "steps": blocks, "kernel_ms_mean": kms / max(1, l),
           "ms_per_block_with_wait_median": float(np.median(w)), "ms_per_block_with_wait_max": float(w.max()), "aux_kernels_ms_per_block": aux_ms / blocks,
           "draws_per_s": 2.0 * V * N * blocks / dt, "finite": bool(torch.isfinite(mix).all().item()),
           "round_4": "the same bank with round 4's library (rand() on the host + a device round trip per block): 5.14 ms per block at 1,024 voices, 82 ms at 16,384 (profiles/r05/noise_bench_ab_first.jsonl, same box)"}
    bank.close()
    return res


def run_realtime(patch, voices, N, blocks=2000, name="realtime_deadline"):
    """max real-time voice count: block time <= N / 48000 s over >= 2000 consecutive blocks, p99, every block synchronised like a
    real-time host would; sustain (2000 blocks) and the worst case (every voice in its release ramp) separately"""
    import torch
    import klang_amd
    notes = NOTES.get(patch, 32)
    base = 1 << 20
    tmp = klang_amd.SynthBank(patch, synths=base // notes, notes=notes, max_block=N)
    rng = np.random.default_rng(7)
    pitches = rng.integers(36, 97, size=base).astype(np.int32)
    rec = tmp.note_records((np.arange(base) // notes).astype(np.int32), pitches, np.full(base, 0.8, np.float32))
    tmp.close()
    bank = klang_amd.SynthBank(patch, synths=voices // notes, notes=notes, max_block=N)
    V = bank.voices
    script = klang_amd.EventScript(bank, 2)
    first = script.add_records(rec)
    v = np.arange(V, dtype=np.int64)
    script.note_on(np.zeros(V, np.int64), v, first + (v % base))               # block 0: every voice starts
    script.note_off(np.ones(V, np.int64), v)                                   # block 1 (played later): every voice is released
    script.commit()
    mix = torch.zeros((2, N), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ts = torch.cuda.Stream()
    torch.cuda.set_stream(ts)
    st = ts.cuda_stream
    mix.zero_(); script.play_device(0, mix.data_ptr(), N, st); torch.cuda.synchronize()
    import gc
    gc_was = gc.isenabled()
    gc.collect(); gc.disable()                                                 # (the interpreter's collector must not run inside a timed block)
    try:
        t = np.empty(blocks)
        for b in range(blocks):
            t0 = time.perf_counter()
            mix.zero_(); bank.process_device(mix.data_ptr(), N, st); torch.cuda.synchronize()
            t[b] = 1e3 * (time.perf_counter() - t0)
        mix.zero_(); script.play_device(1, mix.data_ptr(), N, st); torch.cuda.synchronize()
        r = np.empty(44)
        for b in range(44):
            t0 = time.perf_counter()
            mix.zero_(); bank.process_device(mix.data_ptr(), N, st); torch.cuda.synchronize()
            r[b] = 1e3 * (time.perf_counter() - t0)
    finally:
        if gc_was:
            gc.enable()
    deadline = 1e3 * N / 48000.0
    order = np.argsort(-t)[:10]
    res = {"name": name, "workload": f"{patch}: {V} voices, {blocks} consecutive blocks of {N} samples, each block synchronised (host wall clock per block)",
           "voices": V, "deadline_ms": deadline, "attack_decay_max_ms": float(t[:22].max()), "sustain_p50_ms": float(np.median(t[30:])), "p99_ms": float(np.percentile(t, 99)), "max_ms": float(t.max()),
           "worst_block": int(order[0]), "ten_largest": [[int(i), float(t[i])] for i in order],
           "release_all_ramping_p99_ms": float(np.percentile(r, 99)), "release_max_ms": float(r.max()),
           "worst_block_ms": float(max(t.max(), r.max())), "worst_block_frac_of_deadline": float(max(t.max(), r.max()) / deadline),
           "every_block_within_90_percent_of_the_deadline": bool(max(t.max(), r.max()) <= 0.9 * deadline),
           "realtime": bool(np.percentile(t, 99) <= deadline and np.percentile(r, 99) <= deadline), "unit": "ms per block", "value": float(np.percentile(t, 99)),
           "host": "this Python process (ctypes call + torch.cuda.synchronize per block, garbage collector off)"}
    script.close(); bank.close()
    del mix
    torch.cuda.empty_cache()
    # the same loop from a host that is not an interpreter (klang_amd/host/klang_deadline.cpp: pinned, memory locked, nothing allocates in the loop)
    exe = os.path.join(ROOT, "klang_amd", "host", "klang_deadline")
    if os.path.exists(exe) and patch in PATCH_ID:
        try:
            p = subprocess.run([exe, "--voices", str(V), "--blocks", str(blocks), "--n", str(N), "--patch", str(PATCH_ID[patch]), "--notes", str(notes)], capture_output=True, text=True, timeout=900)
            res["c_host"] = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else {"error": (p.stderr or p.stdout)[-300:]}
        except Exception as e:                                                # noqa: BLE001
            res["c_host"] = {"error": f"{type(e).__name__}: {e}"}
        c = res["c_host"]
        if "max_ms" in c:                                                     # the claim is about the product's own host loop; the interpreter's figures stay beside it
            res["every_block_within_90_percent_of_the_deadline_python_loop"] = res["every_block_within_90_percent_of_the_deadline"]
            res["every_block_within_90_percent_of_the_deadline"] = bool(c["every_block_within_90_percent_of_the_deadline"])
            res["every_block_within_90_percent_of_the_deadline_source"] = "c_host (klang_deadline: C++ loop over klg_process_device + hipStreamSynchronize, pinned, memory locked)"
    return res


def run_max_realtime(patch, N, start_voices, blocks=2000):
    """deadline.max_realtime_voices: the LARGEST bank whose EVERY block — 2,000 sustaining blocks and the 44 of the all-voices release — ends within its
    5.33 ms, found by the product's own host loop (klang_amd/host/klang_deadline.cpp, pinned, memory locked) run AGAINST THE AUDIO CLOCK (--paced 1: block k is not started
    before k periods): sizes from `start_voices` down in steps of 2 Mi until one passes.  The sizes that failed are reported with what failed in them — blocks over the
    deadline, whether they were the device's, and the buffering depth (in blocks) that would have hidden them."""
    exe = os.path.join(ROOT, "klang_amd", "host", "klang_deadline")
    notes = NOTES.get(patch, 32)
    deadline = 1e3 * N / 48000.0
    tried, best = [], None
    V = start_voices
    while V >= (8 << 20) and len(tried) < 6:
        try:
            p = subprocess.run([exe, "--voices", str(V), "--blocks", str(blocks), "--n", str(N), "--patch", str(PATCH_ID[patch]), "--notes", str(notes), "--paced", "1"], capture_output=True, text=True, timeout=600)
            c = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else {"error": (p.stderr or p.stdout)[-300:]}
        except Exception as e:                                                # noqa: BLE001
            c = {"error": f"{type(e).__name__}: {e}"}
        if "max_ms" not in c:
            tried.append({"voices": V, "error": c.get("error", "?")}); break
        over = sum(1 for (_, ms) in c["ten_largest"] if ms > deadline)
        rec = {"voices": V, "p99_ms": c["p99_ms"], "max_ms": c["max_ms"], "release_max_ms": c["release_max_ms"], "every_block_within_the_deadline": bool(c["every_block_within_the_deadline"]),
               "blocks_over_the_deadline_among_the_ten_largest": over, "worst_block_is": c.get("worst_block_is"), "device_max_ms": c.get("device_max_ms"),
               "buffering_that_hides_the_late_blocks_blocks": (c.get("paced") or {}).get("buffering_with_no_gap_in_this_run_blocks")}
        tried.append(rec)
        if rec["every_block_within_the_deadline"]:
            best = V; break
        V -= 2 << 20
    return {"name": "max_realtime_voices", "voices": best, "deadline_ms": deadline, "blocks": blocks, "tried": tried, "host": "klang_deadline --paced 1 (C++, pinned, memory locked, one block of buffering)",
            "reading": "the largest bank of which every one of 2,044 consecutive blocks, started on the audio clock, ended within its own period on this node in this run"}


# ---------------------------------------------------------------------------------------------------------------------------------
def respawn(args):
    """`python bench.py --gpus N` invoked plainly: one process per GPU, launched here the way the driver would"""
    import socket
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get("KLG_BENCH_ONE_GPU") != "1":
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible (KLG_BENCH_ONE_GPU=1 runs every rank on cuda:0 over gloo: a functional test, not a measurement)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))).returncode)


def in_library(args):
    """`bench.py --gpus N --in-library`: the multi-GPU path a C++ host of the library takes — ONE process, klg_init(ids), every bank sharded over the devices inside the
    library (contiguous ranges of synth instances), one ncclAllReduce (RCCL over xGMI, loaded by the library) of the [2][n] block per klg_process_device — beside the
    one-process-per-GPU arrangement the default mode measures.  Weak scaling: --voices per GPU, all sustaining; K timed blocks between two device-wide waits.
    Launched under torch.distributed.run every rank but 0 leaves at once (the library drives all N devices from rank 0's process)."""
    if int(os.environ.get("RANK", "0")) != 0:
        return
    import torch
    import klang_amd
    N, n = args.gpus, args.block
    have = torch.cuda.device_count()
    one_gpu = os.environ.get("KLG_BENCH_ONE_GPU") == "1"
    if have < N and not one_gpu:
        raise SystemExit(f"bench.py --gpus {N} --in-library: only {have} GPU(s) visible (KLG_BENCH_ONE_GPU=1 puts every shard on cuda:0: a functional test, not a measurement)")
    ids = [0] * N if one_gpu else list(range(N))
    patch, notes = args.patch, NOTES.get(args.patch, 32)
    V = groups_voices(args.voices)
    rec_bank = klang_amd.SynthBank(patch, synths=1, notes=notes, max_block=n)          # (records from a one-instance bank on device 0, before the process is made multi-device)
    rng = np.random.default_rng(20250314)
    base = min(V, 1 << 16)
    rec = rec_bank.note_records(np.zeros(base, np.int32), rng.integers(36, 97, size=base).astype(np.int32), np.full(base, 0.8, np.float32))
    rec_bank.close()

    def timed(bank, mix, st, steps):
        for _ in range(max(args.warmup, 40)):                                            # (past the attack: every voice holding at its sustain level)
            mix.zero_(); bank.process_device(mix.data_ptr(), n, st)
        bank.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            mix.zero_(); bank.process_device(mix.data_ptr(), n, st)
        bank.sync(); torch.cuda.synchronize()
        return time.perf_counter() - t0
    # the N = 1 value of THIS invocation (the same workload on device ids[0] alone, before the process is made multi-device): what the N-device value is divided by
    torch.cuda.set_device(ids[0])
    one = klang_amd.SynthBank(patch, synths=V // notes, notes=notes, max_block=n)
    for c0 in range(0, one.voices, base):
        cnt = min(base, one.voices - c0)
        one.voices_upload(np.arange(c0, c0 + cnt, dtype=np.int32), rec[:cnt])
    mix1 = torch.zeros((2, n), dtype=torch.float32, device="cuda")
    ts1 = torch.cuda.Stream()
    with torch.cuda.stream(ts1):
        dt1 = timed(one, mix1, ts1.cuda_stream, args.steps)
    value_one_gpu = one.voices * n * args.steps / dt1
    one.close()
    klang_amd.init(ids)
    bank = klang_amd.SynthBank(patch, synths=(V // notes) * N, notes=notes, max_block=n)
    total = bank.voices
    for c0 in range(0, total, base):
        cnt = min(base, total - c0)
        bank.voices_upload(np.arange(c0, c0 + cnt, dtype=np.int32), rec[:cnt])
    torch.cuda.set_device(ids[0])
    mix = torch.zeros((2, n), dtype=torch.float32, device="cuda")
    ts = torch.cuda.Stream()
    with torch.cuda.stream(ts):
        st = ts.cuda_stream
        dt = timed(bank, mix, st, args.steps)
        bank.timing_begin()                                                              # a second pass with the kernels' own durations (events attached to every dispatch)
        for _ in range(args.steps):
            mix.zero_(); bank.process_device(mix.data_ptr(), n, st)
        bank.sync(); torch.cuda.synchronize()
        info = bank.multi_info(n, probe_reps=200)
        bank.timing_end()
    # the run proves itself: the communicator's own rank count and the devices the shards sit on must be the N that was asked for
    if not one_gpu and (info["rccl_ranks"] != N or info["distinct_devices"] != N or info["shards"] != N):
        raise SystemExit(f"bench.py --gpus {N} --in-library: the bank has {info['shards']} shard(s) on {info['distinct_devices']} device(s) and its RCCL communicator reports {info['rccl_ranks']} rank(s): not reporting a {N}-GPU number")
    value = total * n * args.steps / dt
    out = {"metric": "voice*samples/s @48kHz Subtractive", "value": value, "unit": "voice*samples/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "mode": "in-library",
           "shards": info["shards"], "rccl_ranks_seen": info["rccl_ranks"], "devices_distinct": info["distinct_devices"], "functional_test_only": bool(one_gpu),
           "allreduce_us_per_block": info["allreduce_us"], "per_rank_kernel_ms": [ms / args.steps for ms in info["per_shard_kernel_ms"]],
           "value_one_gpu_same_invocation": value_one_gpu, "weak_scaling_efficiency": value / (N * value_one_gpu),
           "config": {"workload": f"{patch}: {V} voices/GPU, all sustaining, {n}-sample blocks @48kHz, stereo mix resident on device {ids[0]}", "voices_per_gpu": V, "block": n,
                      "parallelism": f"ONE process, klg_init({ids}): voice-shard x{N} inside libklang_mi355.so + " + ("a device-side add of the shards' blocks (one physical GPU)" if one_gpu else f"one ncclAllReduce (RCCL) of [2][{n}] per block"),
                      "mix_checksum": float(mix.abs().sum().item())}}
    bank.close()
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=375)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--patch", default="sub2a")
    ap.add_argument("--voices", type=int, default=375 * 11264, help="voices per GPU (weak scaling), rounded to 375 groups of whole workgroups; 4,224,000 voices = 338 MB of lane records")
    ap.add_argument("--block", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="headline only (what ranks of an N > 1 run do anyway)")
    ap.add_argument("--realtime-voices", type=int, default=32 << 20, help="the deadline test's bank (p99 over 2,000 blocks) and where the search for deadline.max_realtime_voices — the largest bank whose EVERY block fits, the number to quote — starts; 30 - 32 Mi on the nodes measured so far")
    ap.add_argument("--realtime-margin-voices", type=int, default=28 << 20, help="the deadline test with a margin: every block of 28 Mi voices within 90 %% of the 5.33 ms")
    ap.add_argument("--in-library", action="store_true", help="N > 1 through the C-ABI's own sharding: ONE process, klg_init(ids 0..N-1), one ncclAllReduce of the [2][n] block per klg_process_device inside the library (what a C++ host uses) instead of one process per GPU + torch.distributed")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-fx", default="", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)

    if args.in_library:
        return in_library(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return respawn(args)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
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
        # The run proves itself: how many ranks the COMMUNICATOR carries — a sum of ones through it, not argv — and how many different GPUs they sit on
        # (PCI bus ids gathered through it).  Fewer than --gpus of either: no number is reported (KLG_BENCH_ONE_GPU, the functional test on one GPU, says so in the line).
        ones = torch.ones(1, device="cpu" if one_gpu else "cuda")
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        ident = [None] * world
        props = torch.cuda.get_device_properties(local_rank)
        dist.all_gather_object(ident, (os.uname().nodename, str(getattr(props, "uuid", "")) or str(getattr(props, "pci_bus_id", local_rank)), local_rank))
        devices_distinct = len({(h, u) for (h, u, _) in ident})
        if ranks_seen != args.gpus or (devices_distinct != args.gpus and not one_gpu):
            raise SystemExit(f"bench.py --gpus {args.gpus}: the communicator carries {ranks_seen} rank(s) on {devices_distinct} distinct GPU(s): not reporting a {args.gpus}-GPU number")

    import klang_amd
    patch, N = args.patch, args.block
    notes = NOTES.get(patch, 32)
    V = groups_voices(args.voices)
    # weak scaling: every rank owns `V` voices of the global bank (klang_amd/shard.py: contiguous ranges of synth instances)
    sharded = klang_amd.ShardedSynthBank(patch, (V // notes) * world, notes, fs=48000.0, max_block=N, rank=rank, world=world, device=local_rank)
    bank = sharded.bank
    assert bank.voices == V
    bank.random(rank + 1)
    script, sounding = build_script(bank, SCRIPT_BLOCKS, np.random.default_rng(20250314 + rank), cyclic=True)

    # Throughput mode: blocks are pipelined through a small ring of output buffers, so the (2 KiB, latency-bound) all-reduce
    # of block i overlaps the render of block i+1; a buffer is reused only after its own all-reduce has completed.
    RING = 4
    mixes = [torch.zeros((2, N), dtype=torch.float32, device="cuda") for _ in range(RING)]
    pending = [None] * RING
    torch.cuda.synchronize()
    work_stream = torch.cuda.Stream()            # ONE stream for the clear, the library's launches and the collective (a non-default handle:
    torch.cuda.set_stream(work_stream)           # the library takes it as is; the default stream's handle 0 would mean "the bank's own stream")
    stream = work_stream.cuda_stream
    state = {"i": 0}

    def step(collective=True):
        i = state["i"]
        k = i % RING
        state["i"] += 1
        if pending[k] is not None:
            pending[k].wait()                        # the current stream waits for that buffer's collective (issued RING steps ago)
            pending[k] = None
        mixes[k].zero_()
        script.play_device(i % SCRIPT_BLOCKS, mixes[k].data_ptr(), N, stream)      # this block's note events (from HBM), render, reduce
        if world > 1 and collective:
            pending[k] = dist.all_reduce(mixes[k], async_op=True)                 # ONE RCCL all-reduce of the [2][N] block

    def first_pos(i):
        return i % SCRIPT_BLOCKS

    def drain():
        for k in range(RING):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    # settle (untimed set-up, like loading the voices): one full cycle of the script brings every group to its place in the steady state
    for _ in range(SCRIPT_BLOCKS):
        step()
    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    value_alone = None
    if world > 1:
        # the N = 1 value of THIS invocation: every rank renders K blocks of its own share with no collective (what one GPU does alone, on this node, minutes apart from
        # nothing), a whole number of script cycles later the N-rank region starts from the same place in the script
        dist.barrier(); torch.cuda.synchronize()
        i0 = state["i"]
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(collective=False)
        torch.cuda.synchronize()
        dt_alone = time.perf_counter() - t0
        value_alone = float(sum(int(sounding[(i0 + j) % SCRIPT_BLOCKS]) for j in range(args.steps))) * N / dt_alone
        while state["i"] % SCRIPT_BLOCKS != first_pos(i0):
            step(collective=False)
        drain(); torch.cuda.synchronize()
        dist.barrier()
    torch.cuda.synchronize()
    first_timed = state["i"]
    # N = 1: the K timed blocks are submitted as spans of the script (klg_script_render_device: a span's launches — this block's events from HBM,
    # render, reduce — captured as a hipGraph BEFORE the timed region, replayed inside it): K steps, no host call between them.  N > 1 keeps
    # one call per block: every block ends in an all-reduce that torch.distributed issues.  KLG_BENCH_SPANS=0: one call per block at N = 1 too.
    pieces, span_out = [], None
    if world == 1 and os.environ.get("KLG_BENCH_SPANS", "1") != "0":
        span_out = torch.zeros((args.steps, 2, N), dtype=torch.float32, device="cuda")
        pos, left, off = first_timed % SCRIPT_BLOCKS, args.steps, 0
        while left > 0:                                                       # [first, first + K) of the cyclic script, cut where it wraps
            take = min(left, SCRIPT_BLOCKS - pos)
            pieces.append((pos, take, span_out[off].data_ptr())); pos = (pos + take) % SCRIPT_BLOCKS; left -= take; off += take
        if not all(script.capture_span(f, c, ptr, N, stream) for (f, c, ptr) in pieces):
            pieces = []                                                       # (cannot be captured here: one call per block, as below)
        torch.cuda.synchronize()
    if pieces:
        t0 = time.perf_counter()
        for (f, c, ptr) in pieces:
            script.render_device(f, c, ptr, N, stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        state["i"] += args.steps
        checksum = float(span_out[-1].abs().sum().item())
    else:
        bank.timing_begin()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        drain()                                          # every block of the timed region is fully reduced inside the bracket
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        launches, kernel_ms = bank.timing_end()
        multi = None
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_gpu else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            # the collective by itself: 200 all-reduces of the [2][N] block, each between a pair of events on this rank's stream (after 10 untimed)
            probe = torch.zeros((2, N), dtype=torch.float32, device="cuda")
            e0, e1, ar_ms = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), 0.0
            for r in range(210):
                e0.record()
                dist.all_reduce(probe, async_op=True).wait()
                e1.record(); e1.synchronize()
                if r >= 10:
                    ar_ms += e0.elapsed_time(e1)
            gathered = [None] * world
            dist.all_gather_object(gathered, {"rank": rank, "kernel_ms": kernel_ms / max(1, launches), "value_alone": value_alone, "allreduce_us": 1e3 * ar_ms / 200})
            multi = gathered
        checksum = float(mixes[(state["i"] - 1) % RING].abs().sum().item())
    alive_now = int((bank.stages() != 3).sum())
    expect_alive = int(sounding.alive_after[(state["i"] - 1) % SCRIPT_BLOCKS])
    aux_ms_per_step, pass2_ms_per_step = None, None
    if pieces:
        # The kernels' own durations come from the NEXT K blocks of the same steady state (every block of the cyclic script sees the same population),
        # submitted the same way — as spans of the script — with kernel timing armed: events attached to every dispatch; a span is then one call that
        # launches its blocks back to back instead of a graph replay.  Render kernel: klg_timing_end; the blocks' event kernel and reduce: klg_timing_end_aux.
        # (K' = max(K, 100) blocks, after at least one cycle of the script untimed: the chip idles through the read-backs above, and a short run — the driver's
        #  K = 20 is 8 ms — would be measured while the clocks are still on their way up: 0.42 instead of 0.37 ms per kernel)
        KT, KW = max(args.steps, 100), max(args.steps, SCRIPT_BLOCKS)
        tbuf = torch.zeros((KT, 2, N), dtype=torch.float32, device="cuda")
        def cut(pos, count):
            out, off = [], 0
            while count > 0:
                take = min(count, SCRIPT_BLOCKS - pos, KT - off)
                out.append((pos, take, tbuf[off].data_ptr())); pos = (pos + take) % SCRIPT_BLOCKS; count -= take; off = (off + take) % KT
            return out
        for (f, c, ptr) in cut(state["i"] % SCRIPT_BLOCKS, KW):
            script.render_device(f, c, ptr, N, stream)
        state["i"] += KW
        torch.cuda.synchronize()
        bank.timing_begin()
        t2 = time.perf_counter()
        for (f, c, ptr) in cut(state["i"] % SCRIPT_BLOCKS, KT):
            script.render_device(f, c, ptr, N, stream)
        torch.cuda.synchronize()
        pass2_ms_per_step = 1e3 * (time.perf_counter() - t2) / KT
        aux_launches, aux_ms = bank.timing_end_aux()
        launches, kernel_ms = bank.timing_end()
        aux_ms_per_step = aux_ms / KT
        state["i"] += KT
    sounding_timed = float(sum(int(sounding[(first_timed + j) % SCRIPT_BLOCKS]) for j in range(args.steps)))

    if rank == 0:
        value = world * sounding_timed * N / dt
        ms_per_step = 1e3 * dt / args.steps
        kern_s = 1e-3 * kernel_ms / max(1, launches)
        live_mean = sounding_timed / args.steps
        ab = alg_bytes(patch, bank, live_mean, N)
        achieved = ab / kern_s / 1e9
        flops = FLOPS_PER_VOICE_SAMPLE.get(patch, 0) * live_mean * N / kern_s / 1e12
        kernel_name = KERNEL_OF.get(patch, patch) if os.environ.get("KLG_RENDER_X1") != "1" else "klg_render<klg::PatchSub2a"
        traffic, traffic_how = (None, "N > 1: not collected") if world > 1 else pmc_traffic_live(args, kernel_name)
        out = {
            "metric": "voice*samples/s @48kHz Subtractive",
            "value": value, "unit": "voice*samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            **({"rccl_ranks_seen": ranks_seen, "devices_distinct": devices_distinct, "backend": dist.get_backend(), "functional_test_only": bool(one_gpu),
                "allreduce_us_per_block": max(g["allreduce_us"] for g in multi), "per_rank_kernel_ms": [g["kernel_ms"] for g in sorted(multi, key=lambda g: g["rank"])],
                "value_one_gpu_same_invocation": [g["value_alone"] for g in sorted(multi, key=lambda g: g["rank"])],
                "weak_scaling_efficiency": value / sum(g["value_alone"] for g in multi)} if world > 1 and multi else {}),
            "config": {"workload": f"{patch}: {V} voices/GPU (Saw>>Biquad LPF>>ADSR) playing SURVEY 8(d)'s cfg-2 note script (375 blocks: on at 0, off at 150 + (v mod 64), 0.255 s release) as a steady state of 375 phase-shifted groups; {N}-sample blocks @48kHz; note events applied from an HBM-resident script; stereo mix resident in HBM",
                       "voices_per_gpu": V, "voices_sounding_per_gpu_mean": live_mean, "block": N, "parallelism": f"voice-shard x{world}" + (" + RCCL all-reduce of [2][%d] per block" % N if world > 1 else ""),
                       "value_counts": "sounding voices x samples / s (SURVEY 8d: active voices); resident voices x samples / s = %.6g" % (world * V * N * args.steps / dt),
                       "voices_alive_after_last_block": alive_now, "voices_alive_expected": expect_alive,
                       "block_deadline_ms": 1e3 * N / 48000.0, "mix_checksum": checksum,
                       "submission": ("the K timed blocks as %d span(s) of the script, each captured as a hipGraph before the timed region and replayed in it (klg_script_render_device)" % len(pieces)) if pieces else "one klg_script_play_device call per block"},
            # what BINDS this kernel is fp32 VALU issue (voice state in registers, as north_star prescribes; PMC: VALU busy 82-85 % of the kernel's
            # cycles, profiles/r02_pmc/pmc_sub2a_*.json), so that is the roofline's `bound` and `frac`; the HBM view of the same launch is the `hbm`
            # sub-object (`traffic` = HBM bytes per launch from the PMC counters, as everywhere)
            "roofline": {"bound": "valu", "achieved": flops, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / FP32_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_how,
                         "kernel": kernel_name, "kernel_ms": 1e3 * kern_s,
                         "kernel_ms_source": ("HIP events attached to the dispatches of the next max(K, 100) blocks of the same steady state (after a cycle of the script untimed), submitted as spans like the timed ones (launches back to back instead of a graph replay)" if pieces else "HIP events attached to the dispatches of the timed blocks"),
                         "step_kernels_ms": (1e3 * kern_s + aux_ms_per_step) if aux_ms_per_step is not None else None,
                         "step_kernels": "render + this block's event kernel + the voice-mix reduce (klg_timing_end + klg_timing_end_aux)",
                         "ms_per_step_of_the_timing_pass": pass2_ms_per_step,
                         # (the kernels are timed in a pass of their own, a few hundred blocks after the timed replay: pass-to-pass clock noise is a few tenths of a percent)
                         "step_kernels_over_step": ((1e3 * kern_s + aux_ms_per_step) / ms_per_step if aux_ms_per_step is not None else None),
                         "kernels_fit_step": (bool(1e3 * kern_s + aux_ms_per_step <= 1.005 * ms_per_step) if aux_ms_per_step is not None else None), "kernels_fit_step_tolerance": 0.005,
                         "flops_per_voice_sample_est": FLOPS_PER_VOICE_SAMPLE.get(patch, 0),
                         "peak_note": "157.3 TFLOP/s is the packed-FMA fp32 peak; bit-parity forbids contraction, so separate mul / add reach at most half of it (v_pk_mul_f32 / v_pk_add_f32) — see `valu.sustain_loop.issue_rate_frac_est` for the fraction of VALU ISSUE slots used",
                         "hbm": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": ab,
                                 "algorithmic_bytes": "sounding voices x (80 B record read + 32 B written back) + the [2][256] block; a silent voice costs its 4-byte flag word (not counted)"},
                         "valu": {"achieved_tflops_est": flops, "peak_tflops": FP32_PEAK_TFLOPS, "frac": flops / FP32_PEAK_TFLOPS,
                                  "flops_per_voice_sample_est": FLOPS_PER_VOICE_SAMPLE.get(patch, 0)}},
        }
        if world == 1 and not args.no_configs:
            script.close(); sharded.close(); bank = None
            torch.cuda.empty_cache()
            configs = []

            def leg(fn, *a, **kw):
                try:
                    configs.append(fn(*a, **kw))
                except Exception as e:                                       # noqa: BLE001  a side leg must never cost the bench line
                    configs.append({"name": getattr(fn, "__name__", "leg") + str(a[:2]), "error": f"{type(e).__name__}: {e}"})
                    print(f"bench.py: {fn.__name__}{a[:2]} failed: {e}", file=sys.stderr)
            leg(run_literal_script, "sub2a", V, N, "cfg2_script_as_written_at_headline_size", phases=True)
            leg(run_literal_script, "sub2a", 1024, N, "cfg2_1024_voices")
            leg(run_literal_script, "supersaw", 16384, N, "cfg3_16384_supersaw_voices")
            leg(run_fx, "pingpong", 4096, N, args=args)
            leg(run_fx, "reverb", 4096, N, args=args)
            leg(run_fx, "pingpong", 4096, N, dials={2: 0.5, 3: 0.5}, tag="_vibrato", args=args)    # Scratch / Rate up (six of PingPong.k's eight presets have them up): the LFO's fp64 sine and a controls[1].set() per sample
            leg(run_fx, "pingpong", 4096, N, dials="random7", tag="_random_dials", args=args)   # one instance in seven with random dials: moving smoothers, vibrato, taps inside a chunk
            leg(run_fx, "pingpong", 65536, N, args=args)                             # the same patch at bank scale: where config 4 meets north_star's "≥ 40 % of the HBM roofline" (100 GB of delay lines)
            leg(run_literal_script, "fm4", 131072, N, "cfg5_share_131072_fm4_voices")
            leg(run_literal_script, "fm4", 1 << 20, N, "cfg5_1048576_fm4_one_gpu")   # config 5 WHOLE on one GPU (no 8-GPU node has been offered to this repo): 1 Mi voices of the 4-operator patch, every block against its 5.33 ms
            leg(run_max_realtime, "sub2a", N, args.realtime_voices)
            leg(run_realtime, "sub2a", args.realtime_voices, N)
            # ... and the largest bank (to the nearest 4 Mi voices) whose WORST block — attack, sustain or the all-voices release — stays within 90 % of the deadline
            leg(run_realtime, "sub2a", args.realtime_margin_voices, N, name="realtime_deadline_with_margin")
            leg(run_fx, "pingpong", 4096, N, tag="_block_by_block", per_block=True)   # the same bank handed over one block per call (a real-time host): nothing runs across a block boundary
            leg(run_noise_notes, 16384, N)
            leg(run_fx_recorded, 4096, N)                                          # the generated path (f1) at config 4's size: the recorded PingPong.k, one block per call
            leg(run_fx_recorded, 4096, N, dials="random7", tag="_random_dials")   # ... with taps inside a chunk (tried again in halves and quarters: klg_graph_staged.hpp)
            out["configs"] = configs
            # every leg's roofline in the object the driver keeps: name -> [frac of the bound's peak, kernel ms per block, HBM traffic / algorithmic bytes (PMC) or null,
            # frac on the distinct bytes (Reverb) or null]; the deadline legs and the Noise leg beside them
            legs = {}
            for c in configs:
                r = c.get("roofline")
                if r:
                    ab_leg = r.get("algorithmic_bytes_per_launch") or (r.get("hbm") or {}).get("algorithmic_bytes_per_launch")
                    legs[c["name"]] = [round(r["frac"], 4), round(c.get("kernel_ms_mean", float("nan")), 5), (round(r["traffic"] / ab_leg, 3) if r.get("traffic") and ab_leg else None),
                                       (round(r["frac_on_distinct_bytes"], 4) if "frac_on_distinct_bytes" in r else None)]
            out["roofline"]["legs"] = legs
            out["roofline"]["legs_columns"] = ["frac (of HBM peak for cfg4_*, of fp32 peak otherwise)", "kernel ms per block", "PMC traffic / algorithmic bytes", "frac on distinct bytes"]
            out["deadline"] = {c["name"]: {"voices": c["voices"], "p99_ms": round(c["p99_ms"], 3), "max_ms": round(c["max_ms"], 3), "worst_block": c["worst_block"], "release_max_ms": round(c["release_max_ms"], 3),
                                           "within_90_percent": c["every_block_within_90_percent_of_the_deadline"],
                                           "c_host": {k: c["c_host"].get(k) for k in ("p99_ms", "max_ms", "worst_block", "release_max_ms", "every_block_within_90_percent_of_the_deadline", "blocks_over_90_percent", "of_them_the_devices", "device_max_ms", "error") if k in c.get("c_host", {})}}
                               for c in configs if c.get("name", "").startswith("realtime") and "p99_ms" in c}
            for c in configs:
                if c.get("name") == "max_realtime_voices":
                    out["deadline"]["max_realtime_voices"] = c.get("voices")
                    out["deadline"]["max_realtime_voices_tried"] = [[t.get("voices"), t.get("max_ms"), t.get("blocks_over_the_deadline_among_the_ten_largest"), t.get("buffering_that_hides_the_late_blocks_blocks")] for t in c.get("tried", [])]
                    out["deadline"]["max_realtime_voices_columns"] = ["voices", "largest block ms (deadline %.3f)" % c.get("deadline_ms", 0), "of the ten largest blocks: over the deadline", "blocks of buffering that hide them"]
                if c.get("name") == "cfg5_1048576_fm4_one_gpu" and "roofline" in c:
                    out["deadline"]["cfg5_1048576_fm4_one_gpu"] = {"kernel_ms_mean": c.get("kernel_ms_mean"), "ms_per_block_max": c.get("ms_per_block_max"), "every_block_within_the_deadline": c.get("every_block_within_the_deadline"), "frac_of_fp32_peak": c["roofline"]["frac"]}
            for c in configs:
                if c.get("name", "").startswith("noise_note") and "value" in c:
                    out["noise"] = {"name": c["name"], "ms_per_block_with_wait_median": round(c["ms_per_block_with_wait_median"], 4), "ms_per_block_queued": round(c["ms_per_step"], 4), "kernel_ms": round(c["kernel_ms_mean"], 4), "round_4_ms_per_block": 82.0}
            lit = configs[0]
            if "phases_ms_per_block" in lit:
                out["config"]["value_sustain_only"] = lit["value_sustain_phase"]
                out["config"]["value_all_voices_ramping"] = lit["value_all_ramping_phase"]
                out["config"]["value_script_as_written"] = lit["value"]
                ws = V / 128.0 * N
                out["roofline"]["valu"].update({"sustain_loop": valu_issue(ws, 1e-3 * lit["phases_ms_per_block"]["sustain_only(40..149)"], 31.8)})
        if world == 1 and not args.no_cpu_baseline:
            port = cpu_baseline(patch, N, budget_s=8.0)                      # the C restatement (oracle/klang_oracle.c)
            try:
                ref = cpu_reference(patch, N, budget_s=8.0)                  # the genuine header, where its binary travelled
            except Exception as e:                                          # noqa: BLE001  a baseline must never cost the bench line
                print(f"bench.py: reference baseline unavailable ({e}); reporting the port", file=sys.stderr)
                ref = None
            out["cpu_baseline"] = dict(ref, port_value=port["value"], port_sample=port["sample"]) if ref else port
            if ref:                                                          # the same sample on every physical core the job may use, at once
                try:
                    node = cpu_reference_node(patch, N, ref["blocks"])
                except Exception as e:                                      # noqa: BLE001
                    print(f"bench.py: node baseline unavailable ({e})", file=sys.stderr)
                    node = None
                if node:
                    out["cpu_baseline"]["node"] = node
        # ONE line on stdout, short enough for a log's tail: the headline with every leg's roofline (`roofline.legs`), the deadline and Noise summaries.  The legs in
        # full (`configs`) go to gpurun_out/bench_full.json (KLG_BENCH_FULL names another file; the copies under profiles/ come from there).
        full_path = os.environ.get("KLG_BENCH_FULL") or os.path.join(ROOT, "gpurun_out", "bench_full.json")
        try:
            os.makedirs(os.path.dirname(full_path), exist_ok=True)
            with open(full_path, "w") as f:
                json.dump(out, f)
            out["full_record"] = os.path.relpath(full_path, ROOT)
        except OSError as e:
            print(f"bench.py: could not write {full_path}: {e}", file=sys.stderr)
        compact = {k: v for k, v in out.items() if k != "configs"}
        r = dict(compact["roofline"])
        for k in ("kernel_ms_source", "step_kernels", "peak_note", "valu"):
            r.pop(k, None)
        if "hbm" in r:
            r["hbm"] = {k: r["hbm"][k] for k in ("achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch") if k in r["hbm"]}
        compact["roofline"] = r
        cfg = dict(compact["config"])
        for k in ("value_counts", "submission"):
            cfg.pop(k, None)
        compact["config"] = cfg
        if "cpu_baseline" in compact:
            cb = dict(compact["cpu_baseline"])
            for k in ("port_sample", "binary", "note"):
                cb.pop(k, None)
            if isinstance(cb.get("node"), dict):
                cb["node"] = {k: cb["node"][k] for k in ("value", "unit", "cores", "kind") if k in cb["node"]}
            compact["cpu_baseline"] = cb
        print(json.dumps(compact))
    if world > 1:
        script.close(); sharded.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
