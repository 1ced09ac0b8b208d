#!/usr/bin/env python3
"""oracle/gen_golden_fxexamples.py — TEST INFRASTRUCTURE.  Golden vectors for thirteen shipped example EFFECTS that have no
hand-written kernel: examples/{Gain/{Gain,Pan,RM,Tremolo}, Filtering/{EQ,IIR,WahWah}, Delay/{Echo,Feedback,Reverb},
Modulation/{Flanger,ModDelay,Chorus}}.k run through the genuine reference header (oracle/_ref/ref_fx_*).  They pin the
recorded graph-effect path (klang::gpu::EffectBank + `kind effect` programs): tests/test_gpu_fx_facade.py.

Run from the repo root in the build container:  python oracle/gen_golden_fxexamples.py
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.environ.get("KLG_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")   # KLG_GOLDEN_OUT: regenerate somewhere else (tests/test_golden_regen_cpu.py compares)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from scenario_io import Scenario  # noqa: E402

# name -> per-control (lo, hi) ranges the instances are spread over (inside the Dial ranges of the patch)
FX = {
    "fx_gain": [(0.0, 1.0)],
    "fx_pan": [(0.0, 1.0)],
    "fx_rm": [(1.0, 900.0), (0.0, 1.0)],
    "fx_tremolo": [(0.5, 18.0), (0.0, 1.0)],
    "fx_eq": [(0.0, 1.0), (0.0, 1.0), (0.0, 1.0)],
    "fx_iir": [(0.05, 1.0)],
    "fx_wahwah": [(300.0, 5000.0), (0.8, 8.0), (0.2, 6.0)],
    "fx_echo": [(0.002, 0.02), (0.0, 0.9)],
    "fx_feedback": [(0.002, 0.02), (0.0, 0.85)],
    "fx_flanger": [(0.1, 4.0), (0.5, 9.0)],
    "fx_moddelay": [(0.1, 4.0), (0.1, 0.9)],
    "fx_chorus": [],
    # Delay/Reverb.k: prepare() sets the damping LPF from a control every block (recorded as the per-block prologue),
    # sixteen `param` members (tap times / gains) live in the record, two Delay<192000> (Reverb2.k computes a tap time in DOUBLE from a control, which a recorded fp32 program cannot express)
    "fx_reverb1": [(0.05, 0.5), (0.02, 0.4), (500.0, 5000.0)],
    # data-dependent `if`s in process() (recorded once per outcome and merged into if / else / phi ops): a three-way clipper and a
    # Toggle that selects the gain
    "fx_clipping": [(1.0, 11.0)],
    "fx_mute": [("choice", (0.0, 1.0))],
    # Filtering/Bands.k: two BPFs re-designed every sample from grouped controls
    "fx_bands": [(10.0, 1000.0), (0.1, 10.0), (10.0, 1000.0), (0.1, 10.0)],
    # Filtering/Objects.k: Noise (one libc rand() per sample, ONE sequence shared by the process's instances) through an LPF set in prepare()
    "fx_objects": [(10.0, 1000.0), (0.1, 10.0)],
    # Delay/PingPong.k: Stereo::Delay<192000> (two lines advanced together), stereo::signal arithmetic, cross-fed channels
    "fx_dpingpong": [(0.002, 0.02), (0.3, 0.9), (0.002, 0.02), (0.0, 0.9)],
    # Delay/Patterns.k: `const int p = controls[0];` (a Menu) picks a row of tap times / gains — an int conversion of a control inside
    # process(): recorded as branches over the menu's values, merged into phis of the rows' constants.  Long enough for the 0.25 s tap.
    "fx_patterns": [("choice", (0.0, 1.0, 2.0))],
}
FX["fx_toppingpong"] = [(0.2, 0.9), (0.005, 0.03), (0.0, 0.6), (0.05, 0.9), (0.3, 1.5), (0.005, 0.03)]   # the shipped examples/PingPong.k (config 4): bound to the hand-written kernel, not recorded
FX["fx_topreverb"] = [(0.0, 1.0), (0.0, 1.0), (0.0, 1.0), (0.0, 1.0), (0.3, 1.0), (2.0, 60.0), (0.1, 1.0), (0.05, 1.0), (0.05, 1.0), (0.0, 0.2)]   # the shipped examples/Reverb.k (config 4): bound to the hand-written kernel
# examples/Chorus.k (top level): a Stereo::Effect of two Mono::Modifier channels (each holding a Controls&), five triangle-LFO modulated taps
# per channel set up in the channels' prepare(), recorded as the per-block prologue
FX["fx_topchorus"] = [(0.2, 1.0), (0.05, 1.5), (0.1, 0.5), (1.0, 5.5), (0.1, 0.5), (2.0, 20.0)]
# tests/patches/fx_tape.k (OUR OWN effect): Delay::set in prepare() places the read head once per block, `signal echo = tape` walks it
FX["fx_owntape"] = [(0.003, 0.04), (0.0, 0.9), (400.0, 6000.0)]
# tests/patches/fx_fdn.k (OUR OWN effect): four user Modifiers (drive, Delay, HPF), signals<4> >> a Hadamard Matrix, the rows fed back, every Modifier processed twice per sample
FX["fx_ownfdn"] = [(0.5, 3.0), (20.0, 900.0), (0.02, 0.24), (0.2, 1.0)]
# Delay/Reverb2.k: a tap time computed in DOUBLE from a control — `(controls[1] + 0.01232 * c) * fs` is float + double, double * float, then tap((float)x) — recorded as
# double registers (f2d / dconst / dlow / dadd / dmul / d2f); two Delay<192000> per channel, eight constant taps, a damping LPF set in prepare().  (Added last: the
# scenarios above keep their random draws.)
FX["fx_reverb2"] = [(0.0, 0.5), (0.0, 0.4), (500.0, 5000.0)]
# tests/patches/fx_comb.k (OUR OWN effect, added after everything else): a Delay<0> sized with resize() read with tap(float), Delay::lagrange() on a Delay<2400>
FX["fx_owncomb"] = [(0.6, 19.0), (0.0, 0.85), (0.0, 1.0)]
# Distortion/Functions.k and Distortion/Shaping.k (added after everything else): a plain-`float` C function applied to the signal stream — `hardclip(in * gain) >> out`,
# `Function<float, float> f(softclip); in >> f(distort) >> out` with softclip = tanh(c * x) / tanh(c), the C library's DOUBLE tanh.  Recorded by compiling the patch
# with -DKLANG_GPU_TRACE_FLOAT (include/klang/klang.h: `float` in the patch's own text names the tracing signal)
FX["fx_functions"] = [(1.0, 25.0)]
FX["fx_shaping"] = [(0.001, 5.6)]
# tests/patches/fx_lines.k (OUR OWN effect, added last): an ARRAY of user Modifiers as a member, Delay -> LPF -> gain, a Matrix with zero entries
FX["fx_ownlines"] = [(2.0, 30.0), (300.0, 9000.0), (0.0, 0.55), (0.2, 1.0)]
# examples/Vocoder.k (added last): 22 band-pass pairs, followers, saws; 27 controls (5 dials + 22 METERs that process() feeds: `... >> follower[b] >> controls[5 + b]`); prepare() — Pitch -> Frequency,
# power(), constants to integer powers — stays host code; `_boost(float)` is a plain C function (-DKLANG_GPU_TRACE_FLOAT)
FX["fx_vocoder"] = [(40.0, 80.0), (1.0, 3.5), (0.01, 0.1), (0.01, 0.1), (0.5, 14.0)]
# tests/patches/fx_short.k (OUR OWN effect, added last): two feedback lines read 0.2 .. 120 samples behind the cursor — inside a chunk of the sample-parallel kernel
FX["fx_ownshort"] = [(0.1, 0.9), (0.5, 40.0), (0.0, 0.5), (0.3, 1.6)]      # (the last two — Bend, Length — change mid-run on every other instance)
SHAPE = {"fx_patterns": dict(K=4, blocks=110),       # name -> instances / blocks (default 9 / 24)
         "fx_topreverb": dict(K=9, blocks=64),
         "fx_reverb2": dict(K=9, blocks=40),
         "fx_vocoder": dict(K=5, blocks=24)}         # 8,192 samples: the early reflections arrive after ~2,600, mid[] ~400 later, late[] ~1,000 after that


def draw(rng, lo, hi):
    return float(rng.choice(hi)) if lo == "choice" else float(rng.uniform(lo, hi))



def scenarios():
    rng = np.random.default_rng(20250929)
    out = {}
    for name, ranges in FX.items():
        K, B = SHAPE.get(name, {}).get("K", 9), SHAPE.get(name, {}).get("blocks", 24)
        s = Scenario(patch=name, block=128, blocks=B, instances=K, burst=2200, seed=int(rng.integers(1, 1 << 30)), dump=list(range(B)))
        for k in range(K):
            for c, (lo, hi) in enumerate(ranges):
                s.control(0, k, c, draw(rng, lo, hi))
        for k in range(0, K, 2):                                   # a control change mid-run on some instances
            for c, (lo, hi) in list(enumerate(ranges))[-2:]:
                s.control(9 + c, k, c, draw(rng, lo, hi))
        s.sort()
        out[name] = s
    return out


def main():
    subprocess.run(["make", "-C", HERE, "ref"], check=True)
    for name, s in scenarios().items():
        scn = os.path.join(GOLD, name + ".scn")
        s.save(scn)
        tmp = f"/tmp/_ref_{name}.bin"
        subprocess.run([os.path.join(HERE, "_ref", "ref_" + name), scn, tmp], check=True)
        d = open(tmp, "rb").read()
        magic, K, N, B, CH = (int(x) for x in np.frombuffer(d, np.int32, 5))
        assert magic == 0x58474C4B
        out = np.frombuffer(d, np.float32, B * K * CH * N, 20).reshape(B, K, CH, N)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), out=out)
        print(f"{name}: ok ({os.path.getsize(os.path.join(GOLD, name + '.npz')) // 1024} KiB), peak {np.abs(out).max():.3f}, channels {CH}")


if __name__ == "__main__":
    main()
