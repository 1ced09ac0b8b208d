#!/usr/bin/env python3
"""oracle/gen_golden_examples.py — TEST INFRASTRUCTURE.  Golden vectors for shipped example synths that have NO hand-written
kernel and NO C restatement: examples/Subtractive/{Breakpoint,Ramp,Release,Filter,Expression}.k run through the genuine reference
header (oracle/_ref/ref_ex_*, built by `make -C oracle ref`, build container only).  They pin the recorded-graph path
(include/klang_mi355_graph.h + the DSL facade): tests/test_gpu_facade.py renders the same .k files, compiled unchanged
against include/klang/klang.h, on the GPU and compares with these fixtures.

Run from the repo root in the build container:  python oracle/gen_golden_examples.py
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.environ.get("KLG_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")   # KLG_GOLDEN_OUT: regenerate somewhere else (tests/test_golden_regen_cpu.py compares)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from scenario_io import load_ref_output, Scenario  # noqa: E402


def scenarios():
    rng = np.random.default_rng(20250928)
    out = {}

    def poly(patch, blocks, dump, ctl=(), off_base=6, ctl_events=(), seeded=False, notes=32):
        s = Scenario(patch=patch, block=256, blocks=blocks, synths=1, notes=notes, dump=dump)
        for i, v in ctl:
            s.ctl.append((i, float(np.float32(v))))
        pitches = rng.choice(np.arange(36, 97), size=20, replace=False)
        for k, p in enumerate(pitches):
            s.on(0 if k < 12 else k - 10, 0, int(p), float(rng.uniform(0.25, 1.0)), int(rng.integers(1, 2**31 - 1)) if seeded else -1)
            s.off(off_base + (k % 7), 0, int(p), 0.0)
        for b, i, v in ctl_events:
            s.control(b, 0, i, v)
        s.sort()
        return s

    out["ex_breakpoint"] = poly("ex_breakpoint", 40, [0, 1, 9, 10, 39], ctl=[(0, 0.05), (1, 0.1)], off_base=30, ctl_events=[(3, 0, 0.2), (3, 1, 0.35)])
    out["ex_ramp"] = poly("ex_ramp", 40, [0, 1, 18, 19, 39], ctl=[(0, 0.1)], off_base=30, ctl_events=[(4, 0, 0.45)])
    out["ex_release"] = poly("ex_release", 48, [0, 1, 6, 7, 20, 47], ctl=[(0, 0.002), (1, 0.1), (2, 0.05), (3, 0.12)], ctl_events=[(5, 2, 0.4), (9, 3, 0.03)])
    out["ex_filter"] = poly("ex_filter", 32, [0, 1, 8, 9, 31])
    # Expression.k: per-sample vibrato (osc.set(f) from an LFO whose rate is itself an envelope), three 3/4-point envelopes,
    # swept LPF, random() in on() -> every note-on carries a seed
    out["ex_expression"] = poly("ex_expression", 64, [0, 1, 30, 31, 63], off_base=20, seeded=True)
    # four more shipped examples: 32 Fast::Sine partials in a user Generator (Additive/Saw.k), and AM / FM / FM2 whose carriers and
    # modulators are re-tuned per sample from `carrier.frequency` and the controls
    out["ex_addsaw"] = poly("ex_addsaw", 24, [0, 1, 23], off_base=8)
    out["ex_am"] = poly("ex_am", 32, [0, 1, 9, 31], off_base=8, ctl=[(0, 0.5), (1, 0.5)], ctl_events=[(4, 0, 1.7), (6, 1, 0.9)])
    out["ex_fmmod"] = poly("ex_fmmod", 32, [0, 1, 9, 31], off_base=8, ctl=[(0, 0.5), (1, 0.5)], ctl_events=[(4, 0, 2.1), (6, 1, 6.0)])
    out["ex_fm2"] = poly("ex_fm2", 32, [0, 1, 9, 31], off_base=8, ctl=[(0, 0.5), (1, 0.5), (2, 0.5)], ctl_events=[(4, 1, 3.0), (6, 2, 7.5)])
    # Operators.k: grouped controls, three 3-point operator envelopes, `(.. >> op3) * adsr` (the ADSR becomes op3's amp), HPF
    out["ex_operators"] = poly("ex_operators", 32, [0, 1, 9, 31], off_base=8, ctl=[(0, 1.0), (1, 0.5), (2, 1.0), (3, 4.296), (4, 2.0)], ctl_events=[(5, 1, 2.5), (7, 3, 1.0)])
    # OUR OWN patches (tests/patches/*.k; 16 notes, so the 20 note-ons also exercise voice stealing)
    out["own_basic_mix"] = poly("own_basic_mix", 48, [0, 1, 7, 8, 47], off_base=10, notes=16)
    out["own_filters_f2"] = poly("own_filters_f2", 40, [0, 1, 7, 8, 39], off_base=8, notes=16)
    out["own_modal_follow"] = poly("own_modal_follow", 40, [0, 1, 7, 8, 39], off_base=8, notes=16)
    # Additive/Nyquist.k and Square.k: `if (osc[o].frequency < fs.nyquist) out += osc[o] / h;` per partial and sample — a
    # data-dependent branch per oscillator (the partial is neither summed NOR advanced above Nyquist); high pitches so that it bites
    for nm in ("ex_nyquist", "ex_square"):
        s = Scenario(patch=nm, block=256, blocks=16, synths=1, notes=32, dump=[0, 1, 15])
        for k, p in enumerate(rng.choice(np.arange(60, 120), size=14, replace=False)):
            s.on(0 if k < 8 else k - 6, 0, int(p), float(rng.uniform(0.25, 1.0)), -1)
            s.off(5 + (k % 7), 0, int(p), 0.0)
        s.sort()
        out[nm] = s
    # Additive/Resynthesis.k: six partials weighted by `GAIN[o]->Amplitude` (dB -> linear on the host), user Oscillator::reset()
    out["ex_resynthesis"] = poly("ex_resynthesis", 24, [0, 1, 23], off_base=8)
    out["own_branches"] = poly("own_branches", 40, [0, 1, 7, 8, 39], off_base=8, notes=16, ctl_events=[(6, 0, 0.9), (20, 0, 0.1)])
    out["own_pluck"] = poly("own_pluck", 48, [0, 1, 7, 8, 47], off_base=10, notes=16, ctl_events=[(12, 0, 0.98), (20, 1, 0.6)])
    out["own_sample"] = poly("own_sample", 40, [0, 1, 7, 8, 39], off_base=8, notes=16)
    out["own_wavetable"] = poly("own_wavetable", 40, [0, 1, 7, 8, 39], off_base=8, notes=16)
    # pluck_keep.k: no Delay::clear() in on() — slots are started again after their note ran out (release 45 ms) and stolen while sounding
    # (6 slots, 30 note-ons): every restart finds the line as the previous note left it
    s = Scenario(patch="own_pluck_keep", block=256, blocks=72, synths=1, notes=6, dump=[0, 1, 13, 14, 40, 71])
    for k in range(30):
        p = int(rng.integers(40, 80)); b0 = 2 * k + (k % 3)
        s.on(b0, 0, p, float(rng.uniform(0.4, 1.0)))
        if k % 4 != 3:
            s.off(b0 + 3 + (k % 5), 0, p, 0.0)
    s.sort()
    out["own_pluck_keep"] = s
    # early_return.k: `return` out of branches of process(); the control flips the nested early exit on and off mid-run
    out["own_early_return"] = poly("own_early_return", 40, [0, 1, 7, 8, 39], off_base=8, notes=16, ctl_events=[(6, 0, 0.9), (20, 0, 0.1)])
    # iirn.k: Filters::IIR<2> / IIR<3> (the general recursive filter) as recorded nodes
    out["own_iirn"] = poly("own_iirn", 40, [0, 1, 7, 8, 39], off_base=8, notes=16)
    # noise_note.k: two Noise generators per note: the process-wide rand() sequence is shared by the sounding notes in slot order, so the
    # staggered note-ons / note-offs move every later voice's draws
    out["own_noise_note"] = poly("own_noise_note", 40, [0, 1, 7, 8, 39], off_base=8, notes=16, ctl_events=[(6, 0, 0.6), (20, 0, 0.0)])
    # smooth_note.k: controls[i].smooth() in a Note: the Synth's control is advanced by every sounding note in turn; dial moves mid-run
    # (the chain is still converging while notes start and end around it), then long enough to reach its fixed point
    out["own_smooth_note"] = poly("own_smooth_note", 40, [0, 1, 3, 4, 7, 8, 39], off_base=8, notes=16, ctl=[(0, 1000.0), (1, 0.8)], ctl_events=[(3, 0, 4000.0), (3, 1, 0.3), (7, 0, 300.0), (12, 1, 1.0)])
    # hardsync.k: set(f, phase) / reset() on Fast::OSM, Fast::Sine and Basic::Sine oscillators from inside branches of process()
    out["own_hardsync"] = poly("own_hardsync", 40, [0, 1, 7, 8, 39], off_base=8, notes=16, ctl_events=[(6, 0, 0.7), (20, 0, 0.0)])
    # vibstring.k: Delay::set(time) with a recorded time EVERY sample in a Note (the read head's position and fraction are written back)
    out["own_vibstring"] = poly("own_vibstring", 48, [0, 1, 7, 8, 47], off_base=10, notes=16, ctl_events=[(12, 0, 0.98), (20, 1, 0.03)])
    # TRUE stereo notes (Stereo::Note, klang.h:4721-4733: `out` is {l, r}, `buffer++ += out`): per-voice output per channel.
    # stereo_note.k writes out.l / out.r differently from a pan param set in on() and two controls; synthx_shape.k is the note shape of
    # examples/SynTHX.k (detuned saw partials, each on ONE channel drawn with random() in on() -> seeded note-ons)
    out["own_stereo_note"] = poly("own_stereo_note", 40, [0, 1, 7, 8, 39], off_base=8, notes=16, ctl_events=[(6, 0, 0.9), (20, 1, 0.2)])
    out["own_synthx_shape"] = poly("own_synthx_shape", 40, [0, 1, 7, 8, 39], off_base=8, notes=16, seeded=True, ctl=[(0, 0.02), (1, 0.015)], ctl_events=[(10, 1, 0.3)])
    # (added in round 4, after every earlier draw of `rng`: the older scenarios keep their notes)
    # finish_body.k: `finished()` as a value — stop() + return inside one branch, `!finished()`, `finished() && x < 0`; the second envelope's length follows a control
    out["own_finish_body"] = poly("own_finish_body", 40, [0, 1, 7, 8, 39], off_base=8, notes=16, ctl_events=[(6, 0, 0.9), (20, 0, 0.05)])
    # pwm.k: set(f, phase, duty) on Fast::Pulse / Fast::Saw / Basic::Pulse from inside branches of process()
    out["own_pwm"] = poly("own_pwm", 40, [0, 1, 7, 8, 39], off_base=8, notes=16, ctl_events=[(6, 0, 0.9), (20, 0, 0.1)])
    # (added in round 5) Additive/Inheritance.k: on() points `Additive* osc` at one of three member oscillators by a Menu control, process() is `*osc >> out` — the body
    # depends on HOST state of the note: recorded per note after its events (-DKLANG_GPU_NOTE_VARIANTS).  The menu moves while notes start: three variants sound together
    out["ex_inheritance"] = poly("ex_inheritance", 32, [0, 1, 5, 9, 31], off_base=12, ctl=[(0, 0.0)], ctl_events=[(3, 0, 1.0), (6, 0, 2.0), (8, 0, 0.0), (9, 0, 2.0)])
    # Subtractive/Modular.k (added last): twenty controls; a Menu picks the filter through a host `int` (`switch (filter)` in FLT::process(): -DKLANG_GPU_NOTE_VARIANTS), data-dependent
    # branches on params and on the signal (the Distortion's clipper), `pow(10, x)` / `pow(2, x)` — the C library's DOUBLE pow / exp2 (klg_glibc_pow.hpp) —, `min(20000, signal)` that
    # returns an int.  The menu, the drive and the LFO / MOD amounts (either sign: both branches of their generators) move while notes start
    out["ex_modular"] = poly("ex_modular", 32, [0, 1, 5, 9, 31], off_base=12,
                             ctl=[(3, 0.4), (12, 0.0), (13, 3000.0), (14, 2.0), (15, 4.0), (16, 0.5), (17, -0.6), (18, 5.0), (19, 0.7), (4, 0.1), (7, 0.2), (8, 0.2)],
                             ctl_events=[(3, 12, 1.0), (6, 12, 2.0), (8, 16, -0.4), (9, 17, 0.5), (10, 15, 1.0), (11, 12, 0.0), (14, 13, 800.0)])
    # (round 6, row a16) env_points.k: a seven-point envelope looping over points 2 .. 5 until off() lifts the loop, Rate-mode envelopes (a jump point, a loop, release in Rate mode),
    # Envelope::Points + sequence(), a ten-point lookup envelope read with at(), Operators with a six-point and a Rate-mode envelope, `sweep == Envelope::Release`
    out["own_env_points"] = poly("own_env_points", 56, [0, 1, 7, 8, 20, 55], off_base=8, notes=16, ctl_events=[(6, 0, 0.9), (20, 0, 0.05)])
    # two_types.k: TWO Note types in one Synth (6 Pad slots, then 6 Bell slots): twenty note-ons fill the Pad slots, then the Bell slots, then steal across both types
    out["own_two_types"] = poly("own_two_types", 48, [0, 1, 7, 8, 20, 47], off_base=6, notes=12, ctl_events=[(6, 0, 0.9), (20, 0, 0.05)])
    # leftovers.k (the path's edge): Generators::Wavetables::Sine / Saw, Envelope::Follower::Window<64> (RMS) and <48> (Mean), Envelope::set(new Envelope::Linear())
    out["own_leftovers"] = poly("own_leftovers", 40, [0, 1, 7, 8, 20, 39], off_base=8, notes=16)
    # one voice each: the mix IS that voice, so the GPU result can be compared bit for bit (no summation-order slack)
    solo_ctl = {"ex_breakpoint": [(0, 0.05), (1, 0.1)], "ex_ramp": [(0, 0.1)], "ex_release": [(0, 0.002), (1, 0.1), (2, 0.05), (3, 0.12)],
                "ex_am": [(0, 1.3), (1, 0.8)], "ex_fmmod": [(0, 1.5), (1, 4.0)], "ex_fm2": [(0, 0.7), (1, 3.0), (2, 6.0)],
                "ex_operators": [(0, 1.0), (1, 0.5), (2, 1.0), (3, 4.296), (4, 2.0)], "own_early_return": [(0, 0.9)], "own_smooth_note": [(0, 2500.0), (1, 0.5)], "ex_inheritance": [(0, 2.0)],
                "ex_modular": [(3, 0.3), (12, 2.0), (13, 1500.0), (14, 4.0), (15, 8.0), (16, -0.7), (17, 0.8), (18, 3.0), (19, 0.5), (4, 0.05), (7, 0.1)]}
    for name in list(out):
        src = out[name]
        s = Scenario(patch=src.patch, block=256, blocks=24, synths=1, notes=src.notes, dump=[0, 23])
        for i, v in solo_ctl.get(name, []):
            s.ctl.append((i, float(np.float32(v))))
        s.on(0, 0, 100 if name in ("ex_nyquist", "ex_square") else 57, 0.8, 4242 if name in ("ex_expression", "own_synthx_shape") else -1)
        s.off(14, 0, 100 if name in ("ex_nyquist", "ex_square") else 57, 0.0)
        out[name + "_solo"] = s
    return out


def main():
    subprocess.run(["make", "-C", HERE, "ref"], check=True)
    for name, s in scenarios().items():
        scn = os.path.join(GOLD, name + ".scn")
        s.save(scn)
        tmp = f"/tmp/_ref_{name}.bin"
        subprocess.run([os.path.join(HERE, "_ref", "ref_" + s.patch), scn, tmp], check=True)
        ref = load_ref_output(tmp)
        mix = ref["mix"]
        keep = dict(per_voice=ref["per_voice"], dump=np.asarray(s.dump, dtype=np.int32), stages=ref["stages"],
                    mix_abs_sum=np.abs(mix.astype(np.float64)).sum(axis=(1, 2)))
        if mix.nbytes <= 100 * 1024:
            keep["mix"] = mix
        else:
            keep["mix_dump"] = mix[np.asarray(s.dump)]
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **keep)
        alive = int((ref["stages"] != 3).sum())
        print(f"{name}: ok ({os.path.getsize(os.path.join(GOLD, name + '.npz')) // 1024} KiB), peak {np.abs(ref['per_voice']).max():.3f}, voice-blocks alive {alive}")


if __name__ == "__main__":
    main()
