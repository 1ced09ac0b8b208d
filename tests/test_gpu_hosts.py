"""The host entry points of the DSL facade (b): the programs under tests/hosts/ are the block callbacks of the
reference's JUCE templates and README around ONE effect / synth / note object —
    pingpong.klang::Stereo::Effect::process(buffers)      klang.h:4708-4716   (templates/juce/effect/Source/PluginProcessor.cpp:169-177)
    note.klang::Note::process(buffer) -> bool             klang.h:4295-4303   (README "Usage in a C++ project")
    synth.klang::Synth::process(float*, int)              klang.h:4440-4466   (mono: last sounding note wins, then the Synth's own process())
    synth.klang::Stereo::Synth::process(float**, int)     klang.h:4830-4858
compiled from the SAME source against the genuine reference header (oracle/gen_golden_hosts.py -> tests/golden/host_*.npz) and against
include/klang/klang.h, where the calls end in libklang_mi355.so.  Everything is compared BIT FOR BIT."""
import os
import subprocess

import numpy as np
import pytest

from scenario_io import Scenario, fx_input

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
REFBIN = os.path.join(ROOT, "oracle", "_ref")
OWNBIN = os.path.join(ROOT, "tests", "cpp", "_bin")


def need(path, shipped):
    if os.path.exists(path):
        return path
    if shipped:
        pytest.skip("built only where the reference's .k files exist (build container); the binary travels in oracle/_ref/")
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True)
    return path


def run_effect_host(binary, scn_name, tmp_path, extra=()):
    exe = need(os.path.join(REFBIN, binary), True)
    s = Scenario.load(os.path.join(GOLDEN, scn_name + ".scn"))
    ref = np.load(os.path.join(GOLDEN, scn_name + ".npz"))["out"]
    B, K, CH, N = ref.shape
    t = np.arange(B * N, dtype=np.uint64)
    outs = []
    for k in range(K):
        x = np.stack([fx_input(s.seed, k, c, t, s.burst) for c in range(CH)]).reshape(CH, B, N).transpose(1, 0, 2).astype(np.float32).copy()
        fin, fout = tmp_path / f"in{k}.bin", tmp_path / f"out{k}.bin"
        x.tofile(fin)
        subprocess.run([exe, os.path.join(GOLDEN, scn_name + ".scn"), str(k), str(fin), str(fout), *extra], check=True)
        outs.append(np.fromfile(fout, np.float32).reshape(B, CH, N))
    return np.stack(outs, 1)


def assert_bits(got, ref):
    # bit for bit; a zero may come back with the other sign (a voice's -0.0 is ADDED to the +0.0 of the cleared mix accumulators: +0.0)
    bad = np.argwhere((got.view(np.uint32) != ref.view(np.uint32)) & ~((got == 0) & (ref == 0)))
    assert len(bad) == 0, f"{len(bad)} of {got.size} samples differ, first at {bad[0]}, max abs err {np.abs(got - ref).max()}"
    assert np.abs(got).max() > 0


@pytest.mark.parametrize("how", ["kernel", "recorded"])
def test_effect_template_callback_with_shipped_pingpong_k(tmp_path, how, monkeypatch):
    """The template's callback as written: every control set() from its parameter EVERY block (which resets the controls[1] that PingPong.k
    itself writes per sample), `pingpong.klang::Stereo::Effect::process(buffers)`.  Nine runs of the one-object host, one per scenario instance."""
    if how == "recorded":                                            # the unchanged file recorded as a graph effect instead of tied to klg_fx_pingpong_x:
        monkeypatch.setenv("KLANG_MI355_FORCE_GRAPH", "1")           # the host's set() overwrites the instance's own copy of the control it writes
    got = run_effect_host("facade_host_fx_toppingpong", "fx_toppingpong", tmp_path)
    assert_bits(got, np.load(os.path.join(GOLDEN, "host_fx_toppingpong.npz"))["out"])


@pytest.mark.parametrize("how", ["kernel", "recorded"])
def test_effect_host_reproduces_the_effect_bank_fixture(tmp_path, how, monkeypatch):
    """The same host setting a control only in the block its parameter changes: the fixture of the 9-instance gpu::EffectBank test (fx_toppingpong.npz)."""
    if how == "recorded":
        monkeypatch.setenv("KLANG_MI355_FORCE_GRAPH", "1")
    got = run_effect_host("facade_host_fx_toppingpong", "fx_toppingpong", tmp_path, extra=("--set-on-change",))
    assert_bits(got, np.load(os.path.join(GOLDEN, "fx_toppingpong.npz"))["out"])


@pytest.mark.parametrize("binary,scn,how", [("facade_host_fx_topreverb", "fx_topreverb", "kernel"), ("facade_host_fx_topreverb", "fx_topreverb", "recorded"),
                                            ("facade_host_fx_dpingpong", "fx_dpingpong", "recorded"), ("facade_host_fx_echo", "fx_echo", "recorded"),
                                            ("facade_host_fx_vocoder", "fx_vocoder", "recorded")])      # (27 controls set every block, 22 of them meters process() feeds; prepare() host code on the host's own object)
def test_effect_process_buffer_on_a_host_constructed_object(binary, scn, how, tmp_path, monkeypatch):
    """Reverb.k (tied to its kernel, and RECORDED: KLANG_MI355_FORCE_GRAPH — its prepare() is host code and runs on the host's own object, which
    is the one instance's mirror: gpu::FxRunner::host_prepare), Delay/PingPong.k (Stereo::Effect) and Delay/Echo.k (mono klang::Effect): recorded
    from the construction log of an object the host built itself (`Echo pingpong;`) — no template names the type.  These effects do not write
    their controls, so setting them every block changes nothing: the EffectBank fixtures apply."""
    if how == "recorded" and binary == "facade_host_fx_topreverb":
        monkeypatch.setenv("KLANG_MI355_FORCE_GRAPH", "1")
    got = run_effect_host(binary, scn, tmp_path)
    assert_bits(got, np.load(os.path.join(GOLDEN, scn + ".npz"))["out"])


@pytest.mark.parametrize("which", ["sine", "shaped"])
def test_single_note_process_buffer(which, tmp_path):
    """`note.start(..); if (!note.klang::Note::process(buffer)) note.stop();` on a note with no Synth: overwrite semantics, the bool result,
    release() on the device state, restart after the note ran out."""
    exe = need(os.path.join(OWNBIN, "facade_host_note_" + which), False)
    name = "host_note_" + which
    out = tmp_path / "o.bin"
    subprocess.run([exe, os.path.join(GOLDEN, name + ".scn"), str(out)], check=True)
    ref = np.load(os.path.join(GOLDEN, name + ".npz"))
    B, N = ref["out"].shape
    raw = np.fromfile(out, np.uint8)
    got = raw[:B * N * 4].view(np.float32).reshape(B, N)
    assert np.array_equal(raw[B * N * 4:], ref["finished"]), "Note::process(buffer) -> bool / finished() differ from the reference"
    assert_bits(got, ref["out"])


def test_single_note_equals_config_1_fixture(tmp_path):
    """The same single-note loop on BASELINE config 1 (sine_cfg1: 64 blocks of 1024) equals the per-voice fixture of the Synth-driven run."""
    exe = need(os.path.join(OWNBIN, "facade_host_note_sine"), False)
    out = tmp_path / "o.bin"
    subprocess.run([exe, os.path.join(GOLDEN, "sine_cfg1.scn"), str(out)], check=True)
    ref = np.load(os.path.join(GOLDEN, "sine_cfg1.npz"))
    raw = np.fromfile(out, np.uint8)
    got = raw[:64 * 1024 * 4].view(np.float32).reshape(64, 1024)
    for i, b in enumerate(ref["dump"]):
        assert np.array_equal(got[b].view(np.uint32), ref["per_voice"][i, 0].view(np.uint32)), f"block {b}"


def run_synth_host(exe, scn, tmp_path, env=None):
    out = tmp_path / "s.bin"
    subprocess.run([exe, os.path.join(GOLDEN, scn + ".scn"), str(out)], check=True, env=dict(os.environ, **(env or {})))
    d = open(out, "rb").read()
    magic, N, B, P = (int(x) for x in np.frombuffer(d, np.int32, 4))
    assert magic == 0x4D474C4B
    return np.frombuffer(d, np.float32, B * 2 * N, 16).reshape(B, 2, N), np.frombuffer(d, np.uint8, B * P, 16 + B * 2 * N * 4).reshape(B, P)


@pytest.mark.parametrize("binary,scn,golden", [("facade_host_synth_supersaw", "supersaw_poly", "host_synth_supersaw"), ("facade_host_synth_fm", "fm3_poly", "host_synth_fm"),
                                                 # (notes of different recorded bodies sound together, one bank per body: the block is the bank's that holds the synth's last sounding slot)
                                                 ("facade_host_synth_modular", "ex_modular", "host_synth_modular"), ("facade_host_synth_inheritance", "ex_inheritance", "host_synth_inheritance")])
def test_mono_synth_last_sounding_note_wins_like_the_reference(binary, scn, golden, tmp_path):
    """Shipped SuperSaw.k / FM.k are mono klang::Synths: in the reference their block is the LAST sounding note alone (klang.h:4299).  With
    gpu::LastActiveVoice (here: KLANG_MI355_MONO_MIX=reference) the facade returns exactly that; one voice, so bit for bit."""
    exe = need(os.path.join(REFBIN, binary), True)
    mix, stages = run_synth_host(exe, scn, tmp_path, env={"KLANG_MI355_MONO_MIX": "reference"})
    ref = np.load(os.path.join(GOLDEN, golden + ".npz"))
    assert np.array_equal(stages, ref["stages"])
    assert_bits(mix, ref["mix"])


def test_mono_synth_with_its_own_post_processing(tmp_path):
    """tests/patches/post_synth.k PostMono: last sounding note, then the Synth's own prepare() / process() (LPF set from a control per block,
    gain) over the block — recorded from the Synth's construction log and run as a one-instance effect."""
    exe = need(os.path.join(OWNBIN, "facade_host_synth_postmono"), False)
    mix, stages = run_synth_host(exe, "host_synth_postmono", tmp_path, env={"KLANG_MI355_MONO_MIX": "reference"})
    ref = np.load(os.path.join(GOLDEN, "host_synth_postmono.npz"))
    assert np.array_equal(stages, ref["stages"])
    assert_bits(mix, ref["mix"])


def test_stereo_synth_with_its_own_post_processing(tmp_path):
    """PostStereo: eight voices SUMMED (order of summation differs from the reference's note order: 1e-5), then a different filter per channel."""
    exe = need(os.path.join(OWNBIN, "facade_host_synth_poststereo"), False)
    mix, stages = run_synth_host(exe, "host_synth_poststereo", tmp_path)
    ref = np.load(os.path.join(GOLDEN, "host_synth_poststereo.npz"))
    assert np.array_equal(stages, ref["stages"])
    peak = float(np.abs(ref["mix"]).max())
    assert float(np.abs(mix.astype(np.float64) - ref["mix"]).max()) <= 1e-5 * peak * 4
    assert np.abs(mix).max() > 0
