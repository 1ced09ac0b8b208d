"""The source-compatible DSL façade (include/klang/klang.h): a .k patch's own on()/off() run on the host, blocks are
rendered on the GPU through the C-ABI.  tests/cpp/facade_scenario.cpp is compiled (a) with our tests/patches/sub2a.k and
(b) — in the build container, where the reference exists — with the reference's SHIPPED subtractive.k and SuperSaw.k,
UNCHANGED.  Outputs are compared with the golden mixes / note stages produced by the genuine reference header."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def run_facade(binary, scenario, tmp_path):
    out = tmp_path / "facade.bin"
    subprocess.run([binary, os.path.join(GOLDEN, scenario + ".scn"), str(out)], check=True)
    d = open(out, "rb").read()
    magic, N, B, P = (int(x) for x in np.frombuffer(d, np.int32, 4))
    assert magic == 0x4D474C4B
    mix = np.frombuffer(d, np.float32, B * 2 * N, 16).reshape(B, 2, N)
    stages = np.frombuffer(d, np.uint8, B * P, 16 + B * 2 * N * 4).reshape(B, P)
    return mix, stages


def run_facade_voices(binary, scenario, tmp_path):
    """a driver built with -DDUMP_VOICES: additionally every voice's own block ([dump][P][NC][N]) of the scenario's dump blocks"""
    mix, stages = run_facade(binary, scenario, tmp_path)
    d = open(tmp_path / "facade.bin", "rb").read()
    B, _, N = mix.shape
    P = stages.shape[1]
    o = 16 + B * 2 * N * 4 + B * P
    nd, NC = (int(x) for x in np.frombuffer(d, np.int32, 2, o))
    pv = np.frombuffer(d, np.float32, nd * P * NC * N, o + 8).reshape(nd, P, NC, N)
    return mix, stages, pv


def check(mix, stages, scenario):
    ref = np.load(os.path.join(GOLDEN, scenario + ".npz"))
    assert np.array_equal(stages, ref["stages"]), "note stages (voice allocation / lifecycle) differ from the reference"
    peak = float(np.max(np.abs(ref["per_voice"])))
    V = ref["stages"].shape[1]
    if "mix" in ref:
        err = float(np.max(np.abs(mix.astype(np.float64) - ref["mix"])))
    else:
        err = float(np.max(np.abs(mix[ref["dump"]].astype(np.float64) - ref["mix_dump"])))
    assert err <= 1e-5 * peak * np.sqrt(V) * 4, err
    np.testing.assert_allclose(np.abs(mix.astype(np.float64)).sum(axis=(1, 2)), ref["mix_abs_sum"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("binary,scenario", [("facade_sub2a_n4", "sub2a_steal"), ("facade_sub2a_n4", "sub2a_n64"), ("facade_sub2a_n16", "sub2a_long")])
def test_own_dsl_patch_through_facade(binary, scenario, tmp_path):
    path = os.path.join(ROOT, "tests", "cpp", "_bin", binary)
    if not os.path.exists(path):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True)
    check(*run_facade(path, scenario, tmp_path), scenario)


@pytest.mark.parametrize("binary,scenario", [("facade_sub2b", "sub2b_poly"), ("facade_supersaw", "supersaw_poly"), ("facade_supersaw", "supersaw_ctl")])
def test_shipped_k_files_unchanged_through_facade(binary, scenario, tmp_path):
    path = os.path.join(ROOT, "oracle", "_ref", binary)
    if not os.path.exists(path):
        pytest.skip("built only where the reference's .k files exist (build container); the binary travels in oracle/_ref/")
    check(*run_facade(path, scenario, tmp_path), scenario)


# ---- recorded graph patches (include/klang_mi355_graph.h): the SAME .k files with no KLANG_GPU_BIND line — process() is
# ---- recorded by the facade, compiled for gfx950 with hipRTC at Synth creation, and must match the same goldens
@pytest.mark.parametrize("binary,scenario", [("facade_graph_sub2a_n4", "sub2a_steal"), ("facade_graph_sub2a_n4", "sub2a_n64"), ("facade_graph_sub2a_n16", "sub2a_long")])
def test_own_dsl_patch_recorded_as_graph(binary, scenario, tmp_path):
    path = os.path.join(ROOT, "tests", "cpp", "_bin", binary)
    if not os.path.exists(path):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True)
    check(*run_facade(path, scenario, tmp_path), scenario)


@pytest.mark.parametrize("binary,scenario", [("facade_graph_sub2b", "sub2b_poly"), ("facade_graph_supersaw", "supersaw_poly"), ("facade_graph_supersaw", "supersaw_ctl"),
                                             ("facade_graph_fm", "fm3_poly"), ("facade_graph_fm", "fm3_ctl")])
def test_shipped_k_files_recorded_as_graph(binary, scenario, tmp_path):
    path = os.path.join(ROOT, "oracle", "_ref", binary)
    if not os.path.exists(path):
        pytest.skip("built only where the reference's .k files exist (build container); the binary travels in oracle/_ref/")
    check(*run_facade(path, scenario, tmp_path), scenario)


@pytest.mark.parametrize("name", ["ex_breakpoint", "ex_ramp", "ex_release", "ex_filter", "ex_expression",
                                  "ex_addsaw", "ex_am", "ex_fmmod", "ex_fm2", "ex_operators", "ex_nyquist", "ex_square", "ex_resynthesis",
                                  "ex_inheritance",       # (process() depends on a pointer set in on() — recorded per note, -DKLANG_GPU_NOTE_VARIANTS; three variants sound together)
                                  "ex_modular"])          # (Subtractive/Modular.k: 20 controls, a host int picks the filter, branches on signals, the C library's double pow / exp2, min() returning an int)
def test_example_synths_without_a_handwritten_kernel(name, tmp_path):
    """examples/Subtractive/{Breakpoint,Ramp,Release,Filter,Expression}.k of the reference, compiled unchanged: there is no kernel for
    them in the library, only the recorded graph.  Goldens: oracle/gen_golden_examples.py (genuine reference header)."""
    path = os.path.join(ROOT, "oracle", "_ref", "facade_graph_" + name)
    if not os.path.exists(path):
        pytest.skip("built only where the reference's .k files exist (build container); the binary travels in oracle/_ref/")
    check(*run_facade(path, name, tmp_path), name)


@pytest.mark.parametrize("name", ["own_basic_mix", "own_filters_f2", "own_modal_follow", "own_branches", "own_wavetable", "own_sample", "own_pluck", "own_pluck_keep", "own_early_return", "own_iirn", "own_noise_note", "own_smooth_note", "own_hardsync", "own_vibstring", "own_finish_body", "own_pwm", "own_env_points", "own_two_types", "own_leftovers"])
def test_own_patches_for_the_other_node_kinds(name, tmp_path):
    """tests/patches/{basic_mix,filters_f2,modal_follow}.k (ours): Basic oscillators incl. per-sample set(f), OnePole LPF/HPF, DCF,
    Butterworth<1>/<2>, IIR<1>, Biquad HPF/BPF/BRF/APF, Modal, Envelope::Follower, a member written back by process().
    Fixtures: the same files driven through the genuine header (oracle/gen_golden_examples.py)."""
    path = os.path.join(ROOT, "tests", "cpp", "_bin", "facade_graph_" + name)
    if not os.path.exists(path):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True)
    check(*run_facade(path, name, tmp_path), name)


SOLO = ["ex_breakpoint", "ex_ramp", "ex_release", "ex_filter", "ex_expression", "ex_addsaw", "ex_am", "ex_fmmod", "ex_fm2", "ex_operators", "ex_nyquist", "ex_square", "ex_resynthesis", "ex_inheritance", "ex_modular",
        "own_basic_mix", "own_filters_f2", "own_modal_follow", "own_branches", "own_wavetable", "own_sample", "own_pluck", "own_pluck_keep", "own_early_return", "own_iirn", "own_noise_note", "own_smooth_note", "own_hardsync", "own_vibstring", "own_finish_body", "own_pwm", "own_env_points", "own_two_types", "own_leftovers"]


@pytest.mark.parametrize("name", SOLO)
def test_recorded_graph_single_voice_is_bit_exact(name, tmp_path):
    """One note held and released: the stereo mix is that voice's output, so it must equal the genuine header's BIT FOR BIT."""
    d = os.path.join(ROOT, "oracle", "_ref") if name.startswith("ex_") else os.path.join(ROOT, "tests", "cpp", "_bin")
    path = os.path.join(d, "facade_graph_" + name)
    if not os.path.exists(path):
        if name.startswith("ex_"):
            pytest.skip("built only where the reference's .k files exist (build container); the binary travels in oracle/_ref/")
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True)
    mix, stages = run_facade(path, name + "_solo", tmp_path)
    ref = np.load(os.path.join(GOLDEN, name + "_solo.npz"))
    assert np.array_equal(stages, ref["stages"])
    assert np.array_equal(mix.view(np.uint32), ref["mix"].view(np.uint32)), f"max abs err {np.abs(mix - ref['mix']).max()}"
    assert np.abs(mix).max() > 0


@pytest.mark.parametrize("sp", ["0", "1", "4", "8"])
@pytest.mark.parametrize("binary,scenario", [("facade_graph_sub2a_n4", "sub2a_steal"), ("facade_graph_sub2a_n16", "sub2a_long"), ("facade_graph_fm", "fm3_poly"), ("facade_graph_fm", "fm3_ctl"),
                                             ("facade_graph_supersaw", "supersaw_poly"), ("facade_graph_supersaw", "supersaw_ctl"), ("facade_graph_ex_operators", "ex_operators"),
                                             ("facade_graph_ex_breakpoint", "ex_breakpoint"), ("facade_graph_ex_release", "ex_release"), ("facade_graph_own_two_types", "own_two_types")])
def test_recorded_notes_in_their_sample_parallel_form(binary, scenario, sp, tmp_path, monkeypatch):
    """SURVEY row f1, rank 1 (round 6): small banks of a recorded patch run klg_render_gsp<PatchGen> (klang_amd/csrc/klg_render_sp.hpp) — a tile's samples side by side,
    oscillators closed-form, envelopes and filters walked by the voice's lanes together.  KLG_GRAPH_SP = 0 / 1 / 4 / 8: the voice-per-lane kernel, a voice per wave, four / eight voices
    per wave — the same goldens of the genuine header every way (stages equal, mix within the summation bound; the `_solo` fixtures below are bit for bit)."""
    monkeypatch.setenv("KLG_GRAPH_SP", sp)
    d = os.path.join(ROOT, "tests", "cpp", "_bin") if os.path.exists(os.path.join(ROOT, "tests", "cpp", "_bin", binary)) else os.path.join(ROOT, "oracle", "_ref")
    path = os.path.join(d, binary)
    if not os.path.exists(path):
        pytest.skip("built only where the reference's .k files exist (build container); the binary travels in oracle/_ref/")
    check(*run_facade(path, scenario, tmp_path), scenario)


@pytest.mark.parametrize("sp", ["0", "1", "8"])
@pytest.mark.parametrize("name", ["ex_operators", "ex_breakpoint", "ex_ramp", "ex_release", "ex_fm2", "own_two_types"])
def test_sample_parallel_form_single_voice_is_bit_exact(name, sp, tmp_path, monkeypatch):
    monkeypatch.setenv("KLG_GRAPH_SP", sp)
    test_recorded_graph_single_voice_is_bit_exact(name, tmp_path)


def test_a_body_that_follows_host_state_stops_without_the_variants_switch(tmp_path):
    """examples/Subtractive/Modular.k picks its filter through a host `int` that on() sets from a Menu (`switch (filter)` in process()).  Compiled WITHOUT
    -DKLANG_GPU_NOTE_VARIANTS one body is recorded at notes.add<T>() — round 5 then played that body for every note, silently.  Now every event is followed by a
    recording of the note's process() that is compared with the bank's program: the first note that starts after the menu moved stops the run with the switch's name."""
    exe = os.path.join(ROOT, "oracle", "_ref", "facade_graph_ex_modular_noswitch")
    if not os.path.exists(exe):
        pytest.skip("built only where the reference's .k files exist (build container); the binary travels in oracle/_ref/")
    r = subprocess.run([exe, os.path.join(GOLDEN, "ex_modular.scn"), str(tmp_path / "none.bin")], capture_output=True, text=True)
    assert r.returncode != 0 and "follows HOST state" in r.stderr and "KLANG_GPU_NOTE_VARIANTS" in r.stderr, (r.returncode, r.stderr[-800:])


def test_envelope_record_capacity(tmp_path):
    """an Envelope's lane record holds KLANG_GPU_ENV_POINTS point slots (default 16).  tests/patches/env_points.k assigns seven points in on():
    compiled with 8 slots it renders the same bits from a smaller record; compiled with 6 the note-on STOPS with a message that names the macro — a record never
    holds a truncated envelope (round 5 silently kept the first four points)."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")], check=True)
    b8 = os.path.join(ROOT, "tests", "cpp", "_bin", "facade_graph_own_env_points_cap8")
    mix, stages = run_facade(b8, "own_env_points_solo", tmp_path)
    ref = np.load(os.path.join(GOLDEN, "own_env_points_solo.npz"))
    assert np.array_equal(stages, ref["stages"]) and np.array_equal(mix.view(np.uint32), ref["mix"].view(np.uint32))
    b6 = os.path.join(ROOT, "tests", "cpp", "_bin", "facade_graph_own_env_points_cap6")
    r = subprocess.run([b6, os.path.join(GOLDEN, "own_env_points_solo.scn"), str(tmp_path / "none.bin")], capture_output=True, text=True)
    assert r.returncode != 0 and "holds 7 points" in r.stderr and "KLANG_GPU_ENV_POINTS" in r.stderr, (r.returncode, r.stderr[-600:])


# ---- TRUE stereo notes: Stereo::Note with `out` = {l, r}, `buffer++ += out` (klang.h:4721-4733) ----
STEREO = ["own_stereo_note", "own_synthx_shape"]


def same_bits_or_both_zero(a, b):
    """The fixture's per-voice block is what Stereo::Note::process ADDED to a zeroed buffer (`buffer++ += out`): where `out` is -0.0
    (a note's first sample) the buffer holds 0.0 + -0.0 = +0.0, while klg_process_voices returns `out` itself.  The two zeros add the
    same nothing; every other sample must be the same bits."""
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))))


def stereo_binary(name):
    path = os.path.join(ROOT, "tests", "cpp", "_bin", "facade_graph_" + name)
    if not os.path.exists(path):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True)
    return path


@pytest.mark.parametrize("name", STEREO)
def test_stereo_note_single_voice_is_bit_exact_per_channel(name, tmp_path):
    """One Stereo::Note held and released through Stereo::Synth::process(float**, int): left and right differ, and each equals the genuine
    header's channel BIT FOR BIT — in the mix and in the voice's own [2][n] block (klg_process_voices of a stereo-note bank)."""
    mix, stages, pv = run_facade_voices(stereo_binary(name), name + "_solo", tmp_path)
    ref = np.load(os.path.join(GOLDEN, name + "_solo.npz"))
    assert ref["per_voice"].ndim == 4 and ref["per_voice"].shape[2] == 2
    assert np.array_equal(stages, ref["stages"])
    assert np.array_equal(mix.view(np.uint32), ref["mix"].view(np.uint32)), f"max abs err {np.abs(mix - ref['mix']).max()}"
    assert same_bits_or_both_zero(pv, ref["per_voice"])
    assert np.abs(mix[:, 0]).max() > 0 and np.abs(mix[:, 1]).max() > 0
    assert not np.array_equal(mix[:, 0], mix[:, 1]), "a true stereo note: the channels must differ"


@pytest.mark.parametrize("name", STEREO)
def test_stereo_notes_polyphonic(name, tmp_path):
    """20 note-ons on 16 slots (voice stealing), control changes mid-run: every voice's left and right block is bit-exact against the genuine
    header, the note stages are equal, and each channel of the mix is the sum of ITS channel of the voices (1e-5: summation order)."""
    mix, stages, pv = run_facade_voices(stereo_binary(name), name, tmp_path)
    ref = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert np.array_equal(stages, ref["stages"])
    assert same_bits_or_both_zero(pv, ref["per_voice"]), f"per-voice max abs err {np.abs(pv - ref['per_voice']).max()}"
    peak = float(np.max(np.abs(ref["per_voice"])))
    V = ref["stages"].shape[1]
    want = ref["mix"] if "mix" in ref else ref["mix_dump"]
    got = mix if "mix" in ref else mix[ref["dump"]]
    assert float(np.max(np.abs(got.astype(np.float64) - want))) <= 1e-5 * peak * np.sqrt(V) * 4
    # the channels are mixed separately: left = sum of the voices' left blocks (fp64 model of the dump blocks)
    model = ref["per_voice"].astype(np.float64).sum(axis=1)                       # [dump][2][N]
    assert float(np.max(np.abs(mix[ref["dump"]] - model))) <= 1e-5 * peak * np.sqrt(V) * 4
    assert float(np.max(np.abs(model[:, 0] - model[:, 1]))) > 0.01 * peak
