"""The effect side of the DSL facade: the reference's shipped example EFFECTS (.k files, unchanged) run through
klang::gpu::EffectBank, which records their process() into a `kind effect` graph program (include/klang_mi355_graph.h) —
Delay<192000> members become rings in HBM, controls[i].smooth() a per-instance state word.  Nine instances with different
controls, control changes mid-run, 24 blocks; compared BIT FOR BIT with the genuine header's output
(oracle/gen_golden_fxexamples.py)."""
import os
import subprocess

import numpy as np
import pytest

from scenario_io import Scenario, fx_input

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
NAMES = ["fx_gain", "fx_pan", "fx_rm", "fx_tremolo", "fx_eq", "fx_iir", "fx_wahwah", "fx_echo", "fx_feedback", "fx_flanger", "fx_moddelay", "fx_chorus", "fx_reverb1", "fx_reverb2", "fx_clipping", "fx_mute", "fx_bands", "fx_objects", "fx_dpingpong", "fx_patterns", "fx_topchorus",
         "fx_functions", "fx_shaping",      # (these two: plain-`float` C functions on the signal stream, compiled with -DKLANG_GPU_TRACE_FLOAT — tests/cpp/Makefile)
         "fx_vocoder"]                      # examples/Vocoder.k: 27 controls (22 meters fed by process()), prepare() as host code, power(x, 2.f) in a plain-float function


def run_effect(name, tmp_path, own=False):
    exe = os.path.join(ROOT, "tests", "cpp", "_bin", "facade_" + name) if own else os.path.join(ROOT, "oracle", "_ref", "facade_" + name)
    if own and not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True)
    if not os.path.exists(exe):
        pytest.skip("built only where the reference's .k files exist (build container); the binary travels in oracle/_ref/")
    ref = np.load(os.path.join(GOLDEN, name + ".npz"))["out"]                  # [B][K][CH][N]
    B, K, CH, N = ref.shape
    s = Scenario.load(os.path.join(GOLDEN, name + ".scn"))
    t = np.arange(B * N, dtype=np.uint64)
    x = np.stack([np.stack([fx_input(s.seed, k, c, t, s.burst) for c in range(CH)]) for k in range(K)])    # [K][CH][B*N]
    x = x.reshape(K, CH, B, N).transpose(2, 0, 1, 3).copy()
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    x.astype(np.float32).tofile(fin)
    subprocess.run([exe, os.path.join(GOLDEN, name + ".scn"), str(fin), str(fout)], check=True)
    return np.fromfile(fout, np.float32).reshape(B, K, CH, N), ref


def test_shipped_pingpong_k_bound_to_its_kernel_through_the_effect_bank(tmp_path):
    """examples/PingPong.k (BASELINE config 4), compiled UNCHANGED against the facade with `KLANG_GPU_BIND_FX(PingPong, KLG_PATCH_PINGPONG)`:
    klang::gpu::EffectBank<PingPong> creates the bank with klg_fx_create (the hand-written kernel) instead of recording process().
    Nine instances, control changes mid-run,
    against the genuine header."""
    got, ref = run_effect("fx_toppingpong", tmp_path)
    peak = np.abs(ref).max()
    exact = float((got.view(np.uint32) == ref.view(np.uint32)).mean())
    print(f"bit-exact samples {100 * exact:.2f} %, max abs err {np.abs(got - ref).max():.3e}")
    assert exact == 1.0, f"max abs err {np.abs(got - ref).max()} (peak {peak})"


def test_shipped_pingpong_k_recorded_as_a_graph(tmp_path, monkeypatch):
    """The same unchanged examples/PingPong.k, RECORDED (KLANG_MI355_FORCE_GRAPH=1 ignores the binding): process() writes controls[1]
    twice (`controls[1].set(..)`: the control becomes state of the instance), branches on `std::abs(delay - new_delay) > 0.001` (a double
    comparison, decided exactly on floats), re-phases its LFO with set(rate, pi) on one side of the branch, places both read heads with
    Delay::set(time) every sample and reads them with `right * gain` / `>> left`.  Nine instances, control changes mid-run, against the
    genuine header — bit for bit, like the hand-written kernel."""
    monkeypatch.setenv("KLANG_MI355_FORCE_GRAPH", "1")
    got, ref = run_effect("fx_toppingpong", tmp_path)
    exact = float((got.view(np.uint32) == ref.view(np.uint32)).mean())
    print(f"bit-exact samples {100 * exact:.2f} %, max abs err {np.abs(got - ref).max():.3e}")
    assert exact == 1.0, f"max abs err {np.abs(got - ref).max()} (peak {np.abs(ref).max()})"


def test_shipped_reverb_k_bound_to_its_kernel_through_the_effect_bank(tmp_path):
    """examples/Reverb.k (the other config-4 patch: Stereo::Modifier, Array, Stereo::Bank, signals<4> >> Matrix), compiled UNCHANGED
    against the facade and tied to klg_fx_create(KLG_PATCH_REVERB) with KLANG_GPU_BIND_FX.  Its prepare() re-seeds rand() and redraws the
    tap tables whenever a control changed: the scenario changes controls mid-run."""
    got, ref = run_effect("fx_topreverb", tmp_path)
    peak = np.abs(ref).max()
    exact = float((got.view(np.uint32) == ref.view(np.uint32)).mean())
    print(f"bit-exact samples {100 * exact:.2f} %, max abs err {np.abs(got - ref).max():.3e} (peak {peak:.3f})")
    assert exact == 1.0, f"max abs err {np.abs(got - ref).max()} (peak {peak})"


def test_shipped_reverb_k_recorded_as_a_graph(tmp_path, monkeypatch):
    """The same unchanged examples/Reverb.k, RECORDED (KLANG_MI355_FORCE_GRAPH=1 ignores the binding).  Its prepare() is host code and stays
    host code (it asks Controls::changed(), reseeds rand(), draws the tap tables in a loop over a count, compares caches with !=): the
    EffectBank keeps a host mirror per instance, runs prepare() on it when the instance's dials were set — from the record as the device last
    left it, with every Delay's write cursor told to it — and uploads the words that changed (klg_fx_download_record / klg_fx_upload_words).
    process() is recorded once: Stereo::Modifiers inside Modifiers, `Array<float, 20> times` / `Array<stereo::signal, 20> gains` read from the
    record, `for (d = 0; d < times.count; d++)` as twenty nested recorded branches, Stereo::Bank<LPF / HPF>, sixteen FilteredDelays each processed
    twice per sample, `signals<4> >> Matrix` — ~630 ops, 91 nodes, 273 record words.  Nine instances, dials changed mid-run (a change that
    leaves the early reflections alone draws DIFFERENT random delays for the late ones than the first call did), bit for bit against the
    genuine header — like the hand-written kernel."""
    monkeypatch.setenv("KLANG_MI355_FORCE_GRAPH", "1")
    got, ref = run_effect("fx_topreverb", tmp_path)
    exact = float((got.view(np.uint32) == ref.view(np.uint32)).mean())
    print(f"bit-exact samples {100 * exact:.2f} %, max abs err {np.abs(got - ref).max():.3e} (peak {np.abs(ref).max():.3f})")
    assert exact == 1.0, f"max abs err {np.abs(got - ref).max()}"


@pytest.mark.parametrize("name", NAMES)
def test_example_effect_recorded_as_graph_is_bit_exact(name, tmp_path):
    exe = os.path.join(ROOT, "oracle", "_ref", "facade_" + name)
    if not os.path.exists(exe):
        pytest.skip("built only where the reference's .k files exist (build container); the binary travels in oracle/_ref/")
    ref = np.load(os.path.join(GOLDEN, name + ".npz"))["out"]                  # [B][K][CH][N]
    B, K, CH, N = ref.shape
    s = Scenario.load(os.path.join(GOLDEN, name + ".scn"))
    t = np.arange(B * N, dtype=np.uint64)
    x = np.stack([np.stack([fx_input(s.seed, k, c, t, s.burst) for c in range(CH)]) for k in range(K)])    # [K][CH][B*N]
    x = x.reshape(K, CH, B, N).transpose(2, 0, 1, 3).copy()
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    x.astype(np.float32).tofile(fin)
    subprocess.run([exe, os.path.join(GOLDEN, name + ".scn"), str(fin), str(fout)], check=True)
    got = np.fromfile(fout, np.float32).reshape(B, K, CH, N)
    bad = np.argwhere(got.view(np.uint32) != ref.view(np.uint32))
    assert len(bad) == 0, f"{len(bad)} of {got.size} samples differ, first at {bad[0]}, max abs err {np.abs(got - ref).max()}"
    assert np.abs(got).max() > 0


def test_own_effect_places_its_read_head_in_prepare_and_walks_it(tmp_path):
    """tests/patches/fx_tape.k (ours): `tape.set(time * fs)` in prepare() — recorded as the per-block prologue — and `signal echo = tape` in
    process(): Delay::process reads under the head that set() placed and moves it on.  The head (position, fraction) is state of the
    instance's record from one sample and one block to the next.  Nine instances, dials changed mid-run, against the genuine header."""
    got, ref = run_effect("fx_owntape", tmp_path, own=True)
    exact = float((got.view(np.uint32) == ref.view(np.uint32)).mean())
    print(f"bit-exact samples {100 * exact:.2f} %, max abs err {np.abs(got - ref).max():.3e}")
    assert exact == 1.0 and np.abs(ref).max() > 0.1


def test_own_fdn_effect_with_user_modifiers_signals4_and_matrix(tmp_path):
    """tests/patches/fx_fdn.k (ours): a four-line feedback network of user Modifiers as a recorded effect — each `Tank` is `in * drive >> line >> damp`
    (a Delay<7200> and an HPF), the four read into a `signals<4>`, `taps >> mix` with a Hadamard Matrix (klang.h:1446-1470: rows of products summed left
    to right), `+ in`, every row sent back into its tank — which the `+` chain processes a second time in the same sample; delay times, corner and
    drive follow dials in prepare() (the per-block prologue).  Nine instances, dials changed mid-run, bit for bit against the genuine header."""
    got, ref = run_effect("fx_ownfdn", tmp_path, own=True)
    exact = float((got.view(np.uint32) == ref.view(np.uint32)).mean())
    print(f"bit-exact samples {100 * exact:.2f} %, max abs err {np.abs(got - ref).max():.3e}")
    assert exact == 1.0 and np.abs(ref).max() > 0.1


def test_own_effect_with_an_array_of_user_modifiers_and_a_sparse_matrix(tmp_path):
    """tests/patches/fx_lines.k (ours): `Line line[4]` — an array of a user Modifier type as a member, every element's Delay / LPF / param its own state of the record —
    each line `(in >> tape >> tone) * level` (Delay -> LPF -> gain), set up in a loop in prepare(); `signals<4> >> Matrix` with zero entries (the `0 * x` products are
    computed, as in the reference), the rows fed back.  Nine instances, dials changed mid-run, bit for bit against the genuine header."""
    got, ref = run_effect("fx_ownlines", tmp_path, own=True)
    exact = float((got.view(np.uint32) == ref.view(np.uint32)).mean())
    print(f"bit-exact samples {100 * exact:.2f} %, max abs err {np.abs(got - ref).max():.3e}")
    assert exact == 1.0 and np.abs(ref).max() > 0.1


def test_own_comb_effect_with_a_resizable_delay_and_a_lagrange_tap(tmp_path):
    """tests/patches/fx_comb.k (ours): a `Delay<0>` — the resizable line of klang.h:3515-3624, sized with resize(9600) in the constructor: the node's SIZE is
    read when the recording is finished — as a feedback comb read with tap(float), and `pre.lagrange(t)` (klang.h:3429-3458: four rows, third-order
    weights) on a Delay<2400>.  Nine instances, dials changed mid-run, bit for bit against the genuine header."""
    got, ref = run_effect("fx_owncomb", tmp_path, own=True)
    exact = float((got.view(np.uint32) == ref.view(np.uint32)).mean())
    print(f"bit-exact samples {100 * exact:.2f} %, max abs err {np.abs(got - ref).max():.3e}")
    assert exact == 1.0 and np.abs(ref).max() > 0.1


def test_own_effect_with_taps_inside_the_chunk_of_the_sample_parallel_kernel(tmp_path, monkeypatch, capfd):
    """tests/patches/fx_short.k (ours): two feedback lines read with tap(float) BEFORE the sample's input() at 0.2 .. 120 samples behind the cursor (a smoothed
    dial bent by an LFO) — mostly inside the 32 samples the generated sample-parallel kernel computes side by side: its chunks fail their ring check and are
    tried again in halves and quarters, what no part can take is walked by the plain body, the control path (smoother, LFO) catching up in front of a walk
    (klg_graph_staged.hpp).  Nine instances, dials changed mid-run, bit for bit against the GENUINE header — and the kernel's own counters
    (KLG_FX_STAGED_STAMP=1) say that those ways were taken."""
    import re
    got, ref = run_effect("fx_ownshort", tmp_path, own=True)
    bad = np.argwhere(got.view(np.uint32) != ref.view(np.uint32))
    assert len(bad) == 0, f"{len(bad)} of {got.size} samples differ from the genuine header, first [block, instance, channel, sample] {bad[0]}, max abs err {np.abs(got - ref).max()}"
    assert np.abs(ref).max() > 0.5
    capfd.readouterr()
    monkeypatch.setenv("KLG_FX_STAGED_STAMP", "1")
    got, _ = run_effect("fx_ownshort", tmp_path, own=True)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    rows = re.findall(r"staged parts: (\d+) chunks failed their check, (\d+) parts passed, (\d+) cut in two, (\d+) walked by the plain body, (\d+) control catch-ups", capfd.readouterr().out)
    assert rows, "the recorded effect did not run the sample-parallel kernel"
    total = np.array(rows, dtype=np.int64).sum(axis=0)
    print(f"fx_ownshort over {len(rows)} launches: failed chunks {total[0]}, parts passed {total[1]}, cut in two {total[2]}, plain walks {total[3]}, control catch-ups {total[4]}")
    assert (total[:4] > 0).all(), total
