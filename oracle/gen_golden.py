#!/usr/bin/env python3
"""oracle/gen_golden.py — TEST INFRASTRUCTURE.  Generates tests/golden/* from the GENUINE reference.

Runs only where /root/reference exists (the build container).  It
  1. builds oracle/_ref/* (the reference header + shipped patches compiled where they lie) and
     oracle/_build/* (the C restatement) with oracle/Makefile,
  2. writes seeded note-event scenarios (numpy default_rng(20250314)),
  3. runs the reference binaries on them and stores the results as small fixtures:
        tests/golden/prims.kat          per-primitive known-answer vectors (ref_prims)
        tests/golden/<name>.scn         the scenario (text)
        tests/golden/<name>.npz         per-voice blocks, mix, stages produced by the reference
  4. cross-checks the C restatement bit-for-bit against every fixture (fails loudly otherwise).

Nothing of the reference's source text is stored: fixtures are inputs and outputs only.
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

REF_BIN = {"sine": "ref_sine", "bsine": "ref_sine", "sub2a": "ref_subtractive", "sub2b": "ref_subtractive",
           "supersaw": "ref_supersaw", "fm3": "ref_fm", "fm4": "ref_fm", "pingpong": "ref_pingpong", "reverb": "ref_reverb"}


def synth_scenarios():
    rng = np.random.default_rng(20250314)
    out = {}

    # BASELINE.json config 1: 1-voice Sine, N=1024, 64 blocks
    for patch in ("sine", "bsine"):
        s = Scenario(patch=patch, block=1024, blocks=64, synths=1, notes=1, dump=[0, 1, 31, 63])
        s.on(0, 0, 69, 1.0)
        out[f"{patch}_cfg1"] = s

    def poly(patch, notes, synths, blocks, dump, ctl=(), seeded=False, off_base=8, block=256):
        s = Scenario(patch=patch, block=block, blocks=blocks, synths=synths, notes=notes, dump=dump)
        for i, v in ctl:
            s.ctl.append((i, v))
        for sy in range(synths):
            pitches = rng.choice(np.arange(36, 97), size=notes, replace=False)
            for k, p in enumerate(pitches):
                vel = float(rng.uniform(0.25, 1.0))
                seed = int(rng.integers(1, 2**31 - 1)) if seeded else -1
                s.on(0 if k < notes - 2 else k, sy, int(p), vel, seed)   # last two notes start late
                s.off(off_base + (k % 5), sy, int(p), 0.0)
        s.sort()
        return s

    out["sub2a_poly"] = poly("sub2a", 8, 2, 24, [0, 1, 8, 9, 12, 23])
    out["sub2b_poly"] = poly("sub2b", 32, 1, 24, [0, 1, 8, 9, 23])
    out["supersaw_poly"] = poly("supersaw", 32, 1, 24, [0, 1, 8, 9, 23], seeded=True)
    out["supersaw_ctl"] = poly("supersaw", 32, 1, 12, [0, 11], ctl=[(0, 0.126), (1, 0.615), (2, 1.0)], seeded=True)
    out["fm3_poly"] = poly("fm3", 32, 1, 24, [0, 1, 8, 9, 23])
    out["fm3_ctl"] = poly("fm3", 32, 1, 12, [0, 11], ctl=[(0, 2.5), (1, 4.0), (2, 7.5), (3, 0.01)])
    out["fm4_poly"] = poly("fm4", 32, 1, 24, [0, 1, 8, 9, 23])
    # block-size independence / odd block
    out["sub2a_n64"] = poly("sub2a", 4, 1, 40, [0, 39], block=64, off_base=20)

    # voice stealing (Notes::assign, klang.h:4336-4372): 4 slots, 9 note-ons, some released first
    s = Scenario(patch="sub2a", block=256, blocks=16, synths=1, notes=4, dump=[0, 3, 6, 9, 15])
    seq = [(0, 60), (0, 64), (0, 67), (1, 72)]
    for b, p in seq:
        s.on(b, 0, p, 0.8)
    s.off(2, 0, 64, 0.0)
    s.off(2, 0, 72, 0.0)
    s.on(3, 0, 48, 0.7)     # steals oldest Release (64)
    s.on(4, 0, 50, 0.7)     # steals remaining Release (72)
    s.on(6, 0, 52, 0.7)     # no Release left -> steals oldest playing (60)
    s.on(7, 0, 53, 0.7)
    s.on(9, 0, 55, 0.7)
    s.sort()
    out["sub2a_steal"] = s

    # long run (2 s): envelope fully through attack/decay/sustain/release/off, drift check
    s = poly("sub2a", 16, 1, 375, [0, 150, 160, 200, 374], off_base=150)
    out["sub2a_long"] = s
    return out


def run(cmd):
    subprocess.run(cmd, check=True)


def main():
    os.makedirs(GOLD, exist_ok=True)
    run(["make", "-C", HERE, "ref", "oracle"])
    run([os.path.join(HERE, "_ref", "ref_prims"), os.path.join(GOLD, "prims.kat")])
    run([os.path.join(HERE, "_build", "ko_kat"), "/tmp/_ko.kat"])
    if open("/tmp/_ko.kat", "rb").read() != open(os.path.join(GOLD, "prims.kat"), "rb").read():
        sys.exit("C restatement KATs differ from the reference KATs")

    scenarios = synth_scenarios()
    try:
        from gen_golden_fx import fx_scenarios
        scenarios.update(fx_scenarios())
    except ImportError:
        pass
    for name, s in scenarios.items():
        scn = os.path.join(GOLD, name + ".scn")
        s.save(scn)
        tmp = f"/tmp/_ref_{name}.bin"
        run([os.path.join(HERE, "_ref", REF_BIN[s.patch]), scn, tmp])
        ref = load_ref_output(tmp)
        tmp2 = f"/tmp/_ko_{name}.bin"
        run([os.path.join(HERE, "_build", "ko_run"), scn, tmp2])
        if open(tmp, "rb").read() != open(tmp2, "rb").read():
            sys.exit(f"C restatement differs from the reference on scenario {name}")
        keep = dict(per_voice=ref["per_voice"], dump=np.asarray(s.dump, dtype=np.int32))
        if "mix" in ref:
            mix = ref["mix"]
            keep["stages"] = ref["stages"]
            keep["mix_abs_sum"] = np.abs(mix.astype(np.float64)).sum(axis=(1, 2))
            if mix.nbytes <= 100 * 1024:
                keep["mix"] = mix
            else:
                keep["mix_dump"] = mix[np.asarray(s.dump)]
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **keep)
        print(f"{name}: ok ({os.path.getsize(os.path.join(GOLD, name + '.npz')) // 1024} KiB), restatement bit-exact")


if __name__ == "__main__":
    main()
