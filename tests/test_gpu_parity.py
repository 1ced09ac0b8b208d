"""GPU parity tests proper: the HIP path (through the C-ABI, libklang_mi355.so) against
  (1) the committed golden vectors produced by the genuine reference header, and
  (2) the TEST-ONLY C restatement on larger seeded scenarios,
within the tolerance north_star states: 1e-5 relative (|a-b| <= 1e-5 * max(|ref|, block peak)).
Integer state (stages, voice allocation) must match exactly."""
import glob
import os

import numpy as np
import pytest

from klg_driver import bit_exact_fraction, rel_err, run_scenario_gpu, run_scenario_oracle
from scenario_io import Scenario

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SYNTH_SCENARIOS = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.scn"))
                         if Scenario.load(p).instances == 0 and not os.path.basename(p).startswith(("ex_", "own_", "fx_", "host_")))   # ex_*: facade-only (test_gpu_facade.py)
TOL = 1e-5


@pytest.mark.parametrize("name", SYNTH_SCENARIOS)
def test_golden_scenario(name):
    s = Scenario.load(os.path.join(GOLDEN, name + ".scn"))
    ref = np.load(os.path.join(GOLDEN, name + ".npz"))
    got = run_scenario_gpu(s)
    assert np.array_equal(got["stages"], ref["stages"]), "note stages differ"
    err = rel_err(got["per_voice"], ref["per_voice"])
    frac = bit_exact_fraction(got["per_voice"], ref["per_voice"])
    print(f"{name}: per-voice rel err {err:.3e}, bit-exact samples {100 * frac:.2f}%")
    assert err <= TOL
    # stereo mix: the GPU sums voices in a tree, the reference sequentially: tolerance scaled by voices
    if "mix" in ref:
        mref = ref["mix"]
        mgot = got["mix"]
    else:
        mref = ref["mix_dump"]
        mgot = got["mix"][ref["dump"]]
    peak = max(1e-30, float(np.max(np.abs(ref["per_voice"]))))
    assert float(np.max(np.abs(mgot.astype(np.float64) - mref))) <= TOL * peak * np.sqrt(s.voices) * 4
    np.testing.assert_allclose(np.abs(got["mix"].astype(np.float64)).sum(axis=(1, 2)), ref["mix_abs_sum"], rtol=1e-4, atol=1e-6)


def _poly(patch, synths, notes, blocks, seed, seeded=False, ctl=()):
    rng = np.random.default_rng(seed)
    s = Scenario(patch=patch, block=256, blocks=blocks, synths=synths, notes=notes, dump=list(range(blocks)))
    s.ctl = list(ctl)
    for sy in range(synths):
        for k in range(notes):
            p = int(rng.integers(36, 97))
            s.on(int(rng.integers(0, 2)), sy, p, float(rng.uniform(0.25, 1.0)), int(rng.integers(1, 2**31 - 1)) if seeded else -1)
            if rng.uniform() < 0.5:
                s.off(int(rng.integers(2, blocks)), sy, p, 0.0)
    s.sort()
    return s


@pytest.mark.parametrize("patch,synths,notes,seeded", [
    ("sub2a", 8, 128, False),      # BASELINE config 2: 1024 voices = 8 Stereo::Synth x 128 notes, N = 256
    ("sub2b", 32, 32, False),
    ("supersaw", 32, 32, True),
    ("fm3", 16, 32, False),
    ("fm4", 16, 32, False),
])
def test_against_oracle_1024_voices(patch, synths, notes, seeded, oracle_build):
    s = _poly(patch, synths, notes, 6, seed=1234, seeded=seeded)
    ref = run_scenario_oracle(s, oracle_build)
    got = run_scenario_gpu(s)
    assert np.array_equal(got["stages"], ref["stages"])
    err = rel_err(got["per_voice"], ref["per_voice"])
    print(f"{patch}: {s.voices} voices, rel err {err:.3e}, bit-exact {100 * bit_exact_fraction(got['per_voice'], ref['per_voice']):.2f}%")
    assert err <= TOL


def test_block_size_independence():
    """Size-independent property: rendering 4 x 64 samples equals rendering 1 x 256 (bit-for-bit on the GPU)."""
    def render(block, blocks):
        s = Scenario(patch="sub2a", block=block, blocks=blocks, synths=2, notes=64, dump=list(range(blocks)))
        rng = np.random.default_rng(7)
        for sy in range(2):
            for k in range(64):
                s.on(0, sy, int(rng.integers(36, 97)), 0.8)
        return run_scenario_gpu(s)["per_voice"]
    a = render(256, 2)                 # [2][V][256]
    b = render(64, 8)                  # [8][V][64]
    a = a.transpose(1, 0, 2).reshape(a.shape[1], -1)
    b = b.transpose(1, 0, 2).reshape(b.shape[1], -1)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_mix_is_sum_of_voices_and_accumulates():
    s = _poly("sub2a", 4, 128, 3, seed=99)
    got = run_scenario_gpu(s)
    pv = got["per_voice"].astype(np.float64).sum(axis=1)            # [B][N]
    peak = float(np.max(np.abs(got["per_voice"])))
    assert np.max(np.abs(got["mix"][:, 0, :] - pv)) <= 1e-5 * peak * np.sqrt(s.voices) * 4
    assert np.array_equal(got["mix"][:, 0, :], got["mix"][:, 1, :])   # Mono::Note: L += out; R += out


def test_large_bank_linearity():
    """Full-size property (131072 voices, the 1M/8 per-GPU share): the bank mix equals 1024 x the mix of a
    128-voice bank when every group of 128 voices plays the same notes."""
    import klang_amd
    N = 256
    def run(synths):
        bank = klang_amd.SynthBank("sub2a", synths=synths, notes=128, fs=48000.0, max_block=N)
        rng = np.random.default_rng(5)
        pitches = rng.integers(36, 97, size=128)
        for sy in range(synths):
            for p in pitches:
                bank.note_on(sy, int(p), 0.7)
        out = np.zeros((2, N), np.float32)
        for _ in range(3):
            out[:] = 0
            bank.process(out)
        bank.close()
        return out.astype(np.float64)
    small, big = run(1), run(1024)
    scale = np.max(np.abs(small)) * 1024
    assert np.max(np.abs(big - 1024 * small)) <= 2e-4 * scale


def test_voice_state_roundtrip_and_abi_errors():
    import klang_amd
    bank = klang_amd.SynthBank("sub2a", synths=1, notes=4, max_block=64)
    bank.note_on(0, 60, 1.0)
    out = np.zeros((2, 64), np.float32)
    bank.process(out)
    w = bank.voice_download(0)
    assert w.nbytes == bank.state_bytes == 80 and (w[0] & 3) == 1
    bank.voice_upload(1, w)                                   # clone voice 0 into slot 1
    pv, _ = bank.process_voices(64)
    assert np.array_equal(pv[0].view(np.uint32), pv[1].view(np.uint32)) and np.any(pv[0] != 0)
    with pytest.raises(klang_amd.KlangError):
        bank.process(np.zeros((2, 128), np.float32))          # n > max_block
    with pytest.raises(klang_amd.KlangError):
        bank.note_on(5, 60, 1.0)                              # synth index out of range
    with pytest.raises(klang_amd.KlangError):
        klang_amd.SynthBank("sub2a", synths=1, notes=129)     # Array<NOTE*,128>
    bank.close()


@pytest.mark.parametrize("kernel", ["voice per lane", "two voices per lane", "voice per wave"])
@pytest.mark.parametrize("name", ["sub2a_poly", "sub2a_steal", "sub2a_long"])
def test_every_sub2a_kernel_matches(name, kernel, monkeypatch):
    """Config 2a has three kernels: banks of up to 2,048 voices run one voice per WAVE with the samples side by side (klg_render_sub2a_sp.hpp: what these
    fixtures' banks get by default), larger ones the two-voices-per-lane packed-fp32 kernel (klg_render_x2.hpp); KLG_SUB2A_SP=0 / 1 forces that choice,
    KLG_RENDER_X1=1 selects the generic one-voice-per-lane kernel.  All must reproduce the reference bit for bit."""
    monkeypatch.setenv("KLG_SUB2A_SP", "1" if kernel == "voice per wave" else "0")
    if kernel == "voice per lane":
        monkeypatch.setenv("KLG_RENDER_X1", "1")
    s = Scenario.load(os.path.join(GOLDEN, name + ".scn"))
    ref = np.load(os.path.join(GOLDEN, name + ".npz"))
    got = run_scenario_gpu(s)
    assert np.array_equal(got["stages"], ref["stages"])
    assert bit_exact_fraction(got["per_voice"], ref["per_voice"]) == 1.0


@pytest.mark.parametrize("name", ["supersaw_ctl", "supersaw_poly"])
@pytest.mark.parametrize("lanes", ["0", "1", "2:1", "2:2", "2:4", "3", "3/4", "3/2"])      # ("3/4", "3/2": the sample-parallel kernel with four / two voices per wave, KLG_SUPERSAW_VPW)
def test_all_supersaw_kernels_match(name, lanes, monkeypatch):
    """SuperSaw banks of up to 131,072 voices run the oscillator-pair-per-lane kernel (klg_render_lanes.hpp; 1: its one-oscillator-per-lane
    predecessor), larger ones the voice-per-lane kernel; KLG_SUPERSAW_LANES forces the choice, KLG_SUPERSAW_PAIRS_P the pair kernel's
    sample slots per voice.  All must reproduce the reference bit for bit per voice."""
    if not os.path.exists(os.path.join(GOLDEN, name + ".scn")):
        pytest.skip("no such fixture")
    monkeypatch.setenv("KLG_SUPERSAW_LANES", lanes[0])
    if ":" in lanes:
        monkeypatch.setenv("KLG_SUPERSAW_PAIRS_P", lanes[2])              # sample slots per voice of the pair kernel
    if "/" in lanes:
        monkeypatch.setenv("KLG_SUPERSAW_VPW", lanes[2])
    s = Scenario.load(os.path.join(GOLDEN, name + ".scn"))
    ref = np.load(os.path.join(GOLDEN, name + ".npz"))
    got = run_scenario_gpu(s)
    assert np.array_equal(got["stages"], ref["stages"])
    assert bit_exact_fraction(got["per_voice"], ref["per_voice"]) == 1.0


@pytest.mark.parametrize("patch,synths,notes,seeded", [
    ("sub2a", 4, 128, False), ("sub2b", 8, 32, False), ("supersaw", 8, 32, True), ("fm3", 8, 32, False), ("fm4", 8, 32, False),
])
def test_against_oracle_through_whole_note_lives(patch, synths, notes, seeded, oracle_build):
    """48 blocks (0.26 s) of notes that start at different blocks and are released at random ones: attack and decay ends, release ends and
    operator-envelope segment ends fall INSIDE chunks everywhere — what the event-free chunk paths (env_safe / the packed kernel's bare
    step) must hand to the full Envelope::process at exactly the right sample.  Per voice, against the C restatement."""
    rng = np.random.default_rng(4321)
    blocks = 48
    s = Scenario(patch=patch, block=256, blocks=blocks, synths=synths, notes=notes, dump=list(range(blocks)))
    for sy in range(synths):
        for k in range(notes):
            p = int(rng.integers(36, 97))
            on = int(rng.integers(0, 20))
            s.on(on, sy, p, float(rng.uniform(0.25, 1.0)), int(rng.integers(1, 2**31 - 1)) if seeded else -1)
            if rng.uniform() < 0.8:
                s.off(int(rng.integers(on + 1, blocks - 4)), sy, p, 0.0)
    s.sort()
    ref = run_scenario_oracle(s, oracle_build)
    got = run_scenario_gpu(s)
    assert np.array_equal(got["stages"], ref["stages"])
    err = rel_err(got["per_voice"], ref["per_voice"])
    print(f"{patch}: {s.voices} voices x {blocks} blocks, rel err {err:.3e}, bit-exact {100 * bit_exact_fraction(got['per_voice'], ref['per_voice']):.3f}%")
    assert err <= TOL


def test_whole_note_lives_on_the_voice_per_lane_supersaw_kernel_too(oracle_build, monkeypatch):
    monkeypatch.setenv("KLG_SUPERSAW_LANES", "0")
    test_against_oracle_through_whole_note_lives("supersaw", 8, 32, True, oracle_build)


def test_supersaw_pair_kernel_rebuilds_oscillators_of_records_it_cannot_hold(monkeypatch):
    """The pair form assumes what note_on makes: one duty for the two oscillators of a lane, increments and duties of at least 2^-23.
    Hand-made records (klg_voice_upload) may be anything: a wave that holds one takes the scalar table per oscillator.  Same voices, same
    odd records, through the voice-per-lane kernel and the pair kernel: equal bit for bit — samples and final records."""
    import klang_amd
    def run(lanes, p="1", N=256):
        monkeypatch.setenv("KLG_SUPERSAW_LANES", lanes)
        monkeypatch.setenv("KLG_SUPERSAW_PAIRS_P", p)
        bank = klang_amd.SynthBank("supersaw", synths=2, notes=32, max_block=N)
        rng = np.random.default_rng(11)
        for sy in range(2):
            for k in range(20):
                bank.random(1000 + 32 * sy + k)
                bank.note_on(sy, int(rng.integers(36, 97)), float(rng.uniform(0.3, 1.0)))
        bank.process(np.zeros((2, N), np.float32))
        for v, (osc, word, value) in {3: (1, 2, 0x30000000), 17: (4, 2, 0), 21: (6, 2, 300), 40: (2, 0, 200), 41: (5, 2, 0x08000000)}.items():
            w = bank.voice_download(v)                                    # rec::SuperSaw: flags | 7 x (inc offset duty delta) | adsr
            w[1 + 4 * osc + word] = value
            if word == 0:
                w[1 + 4 * osc + 3] = np.float32(0.0).view(np.uint32)      # delta follows the increment (an increment below 2^-23 of a cycle: 0)
            bank.voice_upload(v, w)
        pv = [bank.process_voices(N)[0].copy() for _ in range(4)]
        recs = np.stack([bank.voice_download(v) for v in range(64)])
        bank.close()
        return np.stack(pv), recs
    for n in (256, 37, 1):                                                # whole chunks; a ragged last iteration; a single sample
        a, ra = run("0", N=n)
        for p in ("1", "2", "4", "sp"):                                   # ("sp": the sample-parallel kernel, klg_render_supersaw_sp.hpp — such voices have ALL their samples redone oscillator by oscillator)
            b, rb = run("2", p, N=n) if p != "sp" else run("3", N=n)
            assert np.array_equal(np.isnan(a), np.isnan(b)), (n, p)
            assert np.array_equal(a[~np.isnan(a)].view(np.uint32), b[~np.isnan(b)].view(np.uint32)), (n, p)
            assert np.array_equal(ra, rb), (n, p)
        assert np.abs(a[~np.isnan(a)]).max() > 0


@pytest.mark.parametrize("p", ["1", "2", "4", "sp"])
@pytest.mark.parametrize("n", [256, 37, 1])
def test_supersaw_pair_kernel_block_lengths(p, n, monkeypatch):
    """Ordinary records (note_on), odd block lengths: the pair kernel's sample slots against the voice-per-lane kernel, samples and records."""
    import klang_amd
    def run(lanes):
        monkeypatch.setenv("KLG_SUPERSAW_LANES", "3" if (p == "sp" and lanes == "2") else lanes)
        monkeypatch.setenv("KLG_SUPERSAW_PAIRS_P", p if p != "sp" else "1")
        bank = klang_amd.SynthBank("supersaw", synths=3, notes=32, max_block=256)
        rng = np.random.default_rng(5)
        for sy in range(3):
            for k in range(24):
                bank.random(77 + 32 * sy + k)
                bank.note_on(sy, int(rng.integers(36, 97)), float(rng.uniform(0.3, 1.0)))
        out = []
        for b in range(6):
            if b == 3:
                for sy in range(3):
                    bank.set_control(sy, 1, 0.3)                          # Saw-Tri: the next note_on carries another duty
                    bank.random(5); bank.note_on(sy, 60, 0.9)
            out.append(bank.process_voices(n)[0].copy())
        recs = np.stack([bank.voice_download(v) for v in range(96)])
        bank.close()
        return np.stack(out), recs
    a, ra = run("0")
    b, rb = run("2")
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(ra, rb) and np.abs(a).max() > 0


@pytest.mark.parametrize("patch,voices,env_ref,env_new", [
    ("supersaw", 100032, {"KLG_SUPERSAW_LANES": "0"}, {}),            # above the grid cap (CUs x 8 workgroups of 32 voices): the grid-stride loop; a last group that is part dead
    ("supersaw", 6144, {"KLG_SUPERSAW_LANES": "0"}, {}),              # more than 128 partial rows: the separate klg_reduce instead of the fused combine
    ("sub2a", 2048, {"KLG_SUB2A_SP": "0"}, {"KLG_SUB2A_SP": "1"}),    # the voice-per-wave kernel at the largest bank it is given, against the packed kernel
    ("sub2a", 2016, {"KLG_SUB2A_SP": "0"}, {"KLG_SUB2A_SP": "1"}),
])
def test_sample_parallel_kernels_at_the_sizes_the_small_fixtures_do_not_reach(patch, voices, env_ref, env_new, monkeypatch):
    """klg_render_supersaw_sp is the default at every size and klg_render_sub2a_sp up to 2,048 voices, but the golden fixtures hold ~96 voices.  Here the
    sizes where the launch changes shape: the same notes through the sample-parallel kernel and through the voice-per-lane / packed one — every voice's block, the mix
    and every record afterwards, bit for bit (two thirds of the slots sounding, some released mid-run: live and dead voices share waves and groups)."""
    import klang_amd
    P, N = 32, 256
    def run(env):
        for k in ("KLG_SUPERSAW_LANES", "KLG_SUB2A_SP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        bank = klang_amd.SynthBank(patch, synths=voices // P, notes=P, max_block=N)
        rng = np.random.default_rng(99)
        sy = np.repeat(np.arange(voices // P), 21).astype(np.int32)
        bank.random(4711)
        bank.note_on_many(sy, rng.integers(36, 97, size=sy.size).astype(np.int32), rng.uniform(0.3, 1.0, size=sy.size).astype(np.float32))
        outs = []
        for b in range(3):
            if b == 1:
                off = np.arange(0, voices // P, 3, dtype=np.int32)
                for s_ in off[:200]:
                    bank.note_off(int(s_), int(bank_pitch[int(s_)]))
            pv, mix = bank.process_voices(N)
            outs.append((pv.copy(), mix.copy()))
        recs = np.stack([bank.voice_download(v) for v in list(range(0, 64)) + list(range(voices - 64, voices))])
        stages = bank.stages().copy()
        bank.close()
        return outs, recs, stages
    rngp = np.random.default_rng(99)
    bank_pitch = rngp.integers(36, 97, size=(voices // P) * 21).astype(np.int32)[::21]      # (the first note of every synth: what block 1 releases)
    a, ra, sa = run(env_ref)
    b, rb, sb = run(env_new)
    assert np.array_equal(sa, sb)
    for (pa, ma), (pb, mb) in zip(a, b):
        assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32))
        assert np.allclose(ma, mb, rtol=0, atol=1e-5 * np.sqrt(voices) * 4 * max(1.0, float(np.abs(pa).max())))
        assert np.abs(pa).max() > 0
    assert np.array_equal(ra, rb)
