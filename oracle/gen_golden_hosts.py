"""oracle/gen_golden_hosts.py — TEST INFRASTRUCTURE.  Golden vectors for the host programs of tests/hosts/ (the block callback of the
reference's JUCE effect template; the README's single-note usage), produced by the SAME host sources compiled against the GENUINE
reference header (oracle/_ref/ref_host_*, `make -C oracle ref`; build container only).  Writes tests/golden/host_*.{scn,npz}."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from scenario_io import Scenario, fx_input  # noqa: E402

GOLDEN = os.environ.get("KLG_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")   # KLG_GOLDEN_OUT: regenerate somewhere else (tests/test_golden_regen_cpu.py compares)
REF = os.path.join(ROOT, "oracle", "_ref")


def effect_host(binary, scn_name, out_name, extra=()):
    """every instance of an effect scenario through the single-object host, one run per instance"""
    s = Scenario.load(os.path.join(GOLDEN, scn_name + ".scn"))
    K, B, N = s.instances, s.blocks, s.block
    CH = np.load(os.path.join(GOLDEN, scn_name + ".npz"))["out"].shape[2]
    t = np.arange(B * N, dtype=np.uint64)
    outs = []
    with tempfile.TemporaryDirectory() as d:
        for k in range(K):
            x = np.stack([fx_input(s.seed, k, c, t, s.burst) for c in range(CH)]).reshape(CH, B, N).transpose(1, 0, 2).astype(np.float32).copy()
            x.tofile(os.path.join(d, "in.bin"))
            subprocess.run([os.path.join(REF, binary), os.path.join(GOLDEN, scn_name + ".scn"), str(k), os.path.join(d, "in.bin"), os.path.join(d, "out.bin"), *extra], check=True)
            outs.append(np.fromfile(os.path.join(d, "out.bin"), np.float32).reshape(B, CH, N))
    out = np.stack(outs, 1)                                       # [B][K][CH][N]
    np.savez_compressed(os.path.join(GOLDEN, out_name + ".npz"), out=out)
    print(out_name, out.shape, "peak", float(np.abs(out).max()))
    return out


def note_host(binary, s, name):
    path = os.path.join(GOLDEN, name + ".scn")
    open(path, "w").write(s.text())
    with tempfile.TemporaryDirectory() as d:
        o = os.path.join(d, "o.bin")
        subprocess.run([os.path.join(REF, binary), path, o], check=True)
        raw = np.fromfile(o, np.uint8)
    B, N = s.blocks, s.block
    out = raw[:B * N * 4].view(np.float32).reshape(B, N).copy()
    finished = raw[B * N * 4:].copy()
    assert finished.shape == (B,)
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), out=out, finished=finished)
    print(name, out.shape, "peak", float(np.abs(out).max()), "finished at block", int(np.argmax(finished)) if finished.any() else None)


def synth_host(binary, scn_path, name):
    """the synth's REAL block entry (mono: last sounding note wins + post-processing): stereo block + note stages"""
    with tempfile.TemporaryDirectory() as d:
        o = os.path.join(d, "o.bin")
        subprocess.run([os.path.join(REF, binary), scn_path, o], check=True)
        raw = open(o, "rb").read()
    magic, N, B, P = (int(x) for x in np.frombuffer(raw, np.int32, 4))
    assert magic == 0x4D474C4B
    mix = np.frombuffer(raw, np.float32, B * 2 * N, 16).reshape(B, 2, N).copy()
    stages = np.frombuffer(raw, np.uint8, B * P, 16 + B * 2 * N * 4).reshape(B, P).copy()
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), mix=mix, stages=stages)
    print(name, mix.shape, "peak", float(np.abs(mix).max()), "sounding notes per block (max)", int((stages != 3).sum(1).max()))


def main():
    # the JUCE template sets every control from its parameter EVERY block (PluginProcessor.cpp:173-175); PingPong.k writes controls[1]
    # per sample, so this differs from fx_toppingpong.npz (controls set only when they change) — which the same host reproduces with --set-on-change
    every = effect_host("ref_host_fx_toppingpong", "fx_toppingpong", "host_fx_toppingpong")
    on_change = effect_host("ref_host_fx_toppingpong", "fx_toppingpong", "_tmp_host_check", extra=("--set-on-change",))
    os.remove(os.path.join(GOLDEN, "_tmp_host_check.npz"))
    ref = np.load(os.path.join(GOLDEN, "fx_toppingpong.npz"))["out"]
    assert np.array_equal(on_change.view(np.uint32), ref.view(np.uint32)), "the host with --set-on-change must reproduce fx_toppingpong.npz"
    assert not np.array_equal(every.view(np.uint32), ref.view(np.uint32))

    s = Scenario(patch="host_note_sine", block=256, blocks=10, synths=1, notes=1, dump=[])
    s.on(0, 0, 69, 1.0); s.off(5, 0, 69); s.on(7, 0, 57, 0.5)            # off() = stop(); restarted two blocks later
    note_host("ref_host_note_sine", s, "host_note_sine")
    s = Scenario(patch="host_note_shaped", block=256, blocks=110, synths=1, notes=1, dump=[])
    s.on(0, 0, 57, 0.8); s.off(30, 0, 57); s.on(90, 0, 64, 0.6); s.off(95, 0, 64)   # released, runs out (adsr finished -> stop()), restarted
    note_host("ref_host_note_shaped", s, "host_note_shaped")




    # the mono Synth's own block entry, Synth::process(float*, int): every sounding note overwrites the block in note order (klang.h:4299,
    # 4450-4457) — the shipped SuperSaw.k and FM.k are such Synths — on the polyphonic scenarios of the per-voice fixtures
    synth_host("ref_host_synth_supersaw", os.path.join(GOLDEN, "supersaw_poly.scn"), "host_synth_supersaw")
    synth_host("ref_host_synth_fm", os.path.join(GOLDEN, "fm3_poly.scn"), "host_synth_fm")
    # (round 5) Subtractive/Modular.k and Additive/Inheritance.k are mono Synths too; their notes' bodies follow host state (a Menu control read in on()), so the
    # scenarios of their per-voice fixtures (oracle/gen_golden_examples.py: the menu moves while notes start) have notes of different bodies sounding together
    synth_host("ref_host_synth_modular", os.path.join(GOLDEN, "ex_modular.scn"), "host_synth_modular")
    synth_host("ref_host_synth_inheritance", os.path.join(GOLDEN, "ex_inheritance.scn"), "host_synth_inheritance")
    # Synths with a post-processing process() of their own (tests/patches/post_synth.k), control changes mid-run
    rng = np.random.default_rng(20250928)
    for patch, binary in (("host_synth_postmono", "ref_host_synth_postmono"), ("host_synth_poststereo", "ref_host_synth_poststereo")):
        s = Scenario(patch=patch, block=128, blocks=60, synths=1, notes=8, dump=[])
        for k in range(12):
            b0 = int(rng.integers(0, 30)); p = int(rng.integers(40, 90))
            s.on(b0, 0, p, float(rng.uniform(0.3, 1.0))); s.off(b0 + int(rng.integers(4, 20)), 0, p)
        s.control(10, 0, 0, 4000.0); s.control(25, 0, 1, 0.9); s.control(40, 0, 0, 300.0)
        s.sort()
        path = os.path.join(GOLDEN, patch + ".scn")
        open(path, "w").write(s.text())
        synth_host(binary, path, patch)


if __name__ == "__main__":
    main()
