"""The SAMPLE-PARALLEL form of recorded effects (klang_amd/csrc/klg_graph_staged.hpp): a workgroup takes G instances x C samples of the block at a time,
the recurrences of Effect::process() (klang.h:4208-4216 runs it sample after sample) in sample order on a lane per instance, everything else with a
lane per (sample, instance), the control path a chunk ahead of the audio path.  It must be the SAME function as the one-lane-per-instance walk
(klg_fx_graph<P>) and as the genuine header: every comparison here is bit for bit.

  * the reference's example effects through the facade with KLG_FX_STAGED=0 — the default (staged) run of the same fixtures is
    tests/test_gpu_fx_facade.py; this keeps the other form covered;
  * which of them have a staged form at all, and why the others do not;
  * what the staged form falls back on: taps inside the chunk they are read in (the ring check fails: the chunk is walked by the plain body), ragged
    blocks, blocks shorter than a chunk, dials moved between blocks — recorded PingPong.k against its hand-written kernel and against the one-lane form;
  * recorded Reverb.k, staged against one lane per instance, dials 0 - 4 per instance, odd block lengths.
"""
import os
import re

import numpy as np
import pytest

import klang_amd
from test_gpu_fx_facade import NAMES, run_effect

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("name", NAMES + ["fx_toppingpong", "fx_topreverb"])
def test_example_effect_with_one_lane_per_instance_is_bit_exact(name, tmp_path, monkeypatch):
    monkeypatch.setenv("KLG_FX_STAGED", "0")
    monkeypatch.setenv("KLANG_MI355_FORCE_GRAPH", "1")
    got, ref = run_effect(name, tmp_path)
    bad = np.argwhere(bits(got) != bits(ref))
    assert len(bad) == 0, f"{len(bad)} of {got.size} samples differ, first at {bad[0]}, max abs err {np.abs(got - ref).max()}"


@pytest.mark.parametrize("name", ["fx_owntape", "fx_ownfdn", "fx_owncomb", "fx_ownlines", "fx_ownshort"])
def test_own_effect_with_one_lane_per_instance_is_bit_exact(name, tmp_path, monkeypatch):
    monkeypatch.setenv("KLG_FX_STAGED", "0")
    got, ref = run_effect(name, tmp_path, own=True)
    assert np.array_equal(bits(got), bits(ref))


def test_which_example_effects_have_a_staged_form(tmp_path, monkeypatch, capfd):
    """Every effect of the facade fixtures either gets the staged form or says what it has that the form does not handle.  The config-4 patches and
    the delay effects must have it.  (The program is what the facade prints with KLANG_MI355_DUMP_GRAPH=1 when it creates the bank.)"""
    monkeypatch.setenv("KLANG_MI355_FORCE_GRAPH", "1")
    monkeypatch.setenv("KLANG_MI355_DUMP_GRAPH", "1")
    rows = {}
    for name in NAMES + ["fx_toppingpong", "fx_topreverb"]:
        capfd.readouterr()
        run_effect(name, tmp_path)
        m = re.search(r"^klgg 1\n.*?^end\n", capfd.readouterr().err, re.S | re.M)
        assert m, f"{name}: no program in the facade's dump"
        prog = m.group(0)
        bank = klang_amd.FxBank(prog, 16, max_block=64, channels=2 if "kind effect 2" in prog else 1)
        rows[name] = bank.graph_form()
        bank.close()
    with capfd.disabled():
        for name, f in rows.items():
            print(f"{name:16s} " + (f"staged: {f['instances_per_workgroup']} instances x {f['samples_per_chunk']} samples, {f['levels']} levels, {f['lds_values']} values through LDS" if f["staged"] else "one lane per instance: " + f["why"]))
    for must in ("fx_toppingpong", "fx_topreverb", "fx_echo", "fx_feedback", "fx_dpingpong", "fx_reverb1", "fx_eq", "fx_wahwah"):
        assert rows[must]["staged"], f"{must} lost its staged form: {rows[must]['why']}"
    for f in rows.values():
        assert f["staged"] or f["why"]


def pingpong_program():
    prog = open(os.path.join(GOLDEN, "pingpong_recorded.klgg")).read()
    rec = np.array([int(w, 16) for w in open(os.path.join(GOLDEN, "pingpong_recorded.rec")).read().split()], np.uint32)
    return prog, rec


def test_staged_pingpong_with_near_taps_ragged_blocks_and_moving_dials(monkeypatch):
    """Recorded PingPong.k, three banks fed the same blocks: the hand-written kernel, the staged form, one lane per instance.  70 instances: a third
    with the Delay dial near 0 (the left tap 48 samples behind the cursor, the right one 24: inside a 32-sample chunk — the ring check fails and the
    chunk is tried again in halves and quarters, and what no part can take is walked by the plain body: KLG_FX_STAGED_RETRY), some with vibrato (taps that move), the rest at rest; block lengths that are no multiple of a chunk, shorter
    than one, and 1; dials moved between blocks (host set() on a control the effect itself writes)."""
    prog, rec = pingpong_program()
    K = 70
    rng = np.random.default_rng(11)
    monkeypatch.setenv("KLG_FX_STAGED", "1")
    staged = klang_amd.FxBank(prog, K, max_block=256, initial_record=rec, channels=2)
    form = staged.graph_form()
    assert form["staged"], form["why"]
    monkeypatch.setenv("KLG_FX_STAGED", "0")
    lane = klang_amd.FxBank(prog, K, max_block=256, initial_record=rec, channels=2)
    assert not lane.graph_form()["staged"]
    hand = klang_amd.FxBank("pingpong", K, max_block=256)
    banks = (hand, staged, lane)

    def dial(k, c, v):
        for b in banks:
            b.set_control(k, c, float(v))
    for k in range(K):
        dial(k, 0, rng.uniform(0.2, 0.95))
        if k % 3 == 0:
            dial(k, 5, rng.uniform(0.0, 0.0012)); dial(k, 1, 0.001)          # taps inside the chunk (the right one 0 .. 26 samples behind the cursor: parts of 16 and 8 samples pass where the chunk fails)
            if k % 2 == 0:
                dial(k, 2, rng.uniform(0.6, 1.0)); dial(k, 3, rng.uniform(0.3, 1.0))   # ... that move by the sample (vibrato): a part that passes, then one that does not — the plain body takes over inside a chunk
        elif k % 3 == 1:
            dial(k, 5, rng.uniform(0.05, 0.6)); dial(k, 2, rng.uniform(0.2, 1.0)); dial(k, 3, rng.uniform(0.05, 1.0))   # vibrato
    lengths = [256, 37, 1, 100, 32, 33, 64, 250, 7, 256, 31, 256, 96, 256, 256, 5, 128, 256]
    t = 0
    for bi, n in enumerate(lengths):
        if bi in (4, 9, 13):
            for k in rng.choice(K, 12, replace=False):
                dial(int(k), 5, rng.uniform(0.0, 0.5)); dial(int(k), 1, rng.uniform(0.001, 0.5))
        x = ((rng.random((K, 2, n), dtype=np.float32) - 0.5) * (1.0 if bi < 12 else 0.0)).astype(np.float32)
        outs = [b.process(x.copy()) for b in banks]
        for name, o in zip(("staged", "one lane per instance"), outs[1:]):
            bad = np.argwhere(bits(o) != bits(outs[0]))
            assert len(bad) == 0, f"block {bi} (n = {n}, sample {t}): {name} differs from the hand-written kernel in {len(bad)} samples, first [instance, channel, sample] {bad[0]}"
        t += n
    assert np.abs(outs[0]).max() > 1e-3
    for c in (1, 5):
        for k in (0, 1, 2, 69):
            assert staged.get_control(k, c) == lane.get_control(k, c) == hand.get_control(k, c)
    for b in banks:
        b.close()


@pytest.mark.parametrize("shape", ["16,32", "32,16", "64,8"])
def test_staged_parts_of_a_failed_chunk_with_a_read_head_that_walks(shape, monkeypatch):
    """tests/patches/fx_headcomb.klgg — a feedback comb whose read head is placed once per block (prepare(): delay.set(controls[1]), 1 .. 64 samples) and
    then WALKS with every `delay >> x` (a process() without a set() of its own: the head a part of a chunk starts from is where the part before it left it),
    its sum through a biquad (a serial loop of the audio path, over the part's samples only).  Delays under a chunk fail the chunk's ring check: it is tried
    again in halves and quarters, and what no part can take is walked by the plain body (16 x 32: parts of 16 and 8; 32 x 16: of 8; 64 x 8: no parts) —
    against the same kernel with KLG_FX_STAGED_RETRY=0 (a failed chunk straight to the plain body) and the one-lane-per-instance kernel, bit for bit;
    ragged blocks, dials moved between blocks."""
    prog = open(os.path.join(ROOT, "tests", "patches", "fx_headcomb.klgg")).read()
    g, c = shape.split(",")
    K = 70
    monkeypatch.setenv("KLG_FX_STAGED_G", g); monkeypatch.setenv("KLG_FX_STAGED_C", c)
    banks = []
    for staged, retry in (("1", "1"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("KLG_FX_STAGED", staged); monkeypatch.setenv("KLG_FX_STAGED_RETRY", retry)
        banks.append(klang_amd.FxBank(prog, K, max_block=256, channels=1))
    f = banks[0].graph_form()
    assert f["staged"] and f["instances_per_workgroup"] == int(g) and f["samples_per_chunk"] == int(c), f
    assert banks[1].graph_form()["staged"] and not banks[2].graph_form()["staged"]
    rng = np.random.default_rng(23)
    def dial(k, ctl, v):
        for b in banks:
            b.set_control(k, ctl, float(v))
    for k in range(K):
        dial(k, 0, rng.uniform(0.3, 0.95))
        dial(k, 1, [rng.uniform(1.0, 64.0), float(rng.integers(1, 40)), rng.uniform(8.0, 20.0)][k % 3])
    peak = 0.0
    for bi, n in enumerate([256, 37, 1, 100, 32, 33, 64, 250, 7, 256, 16, 8, 9, 256, 96, 256]):
        if bi in (3, 7, 10, 13):
            for k in rng.choice(K, 20, replace=False):
                dial(int(k), 1, rng.uniform(1.0, 48.0))
        x = (rng.random((K, 1, n), dtype=np.float32) - 0.5).astype(np.float32)
        outs = [b.process(x.copy()) for b in banks]
        for name, o in zip(("a failed chunk straight to the plain body", "one lane per instance"), outs[1:]):
            bad = np.argwhere(bits(o) != bits(outs[0]))
            assert len(bad) == 0, f"block {bi} (n = {n}): the parts differ from '{name}' in {len(bad)} samples, first [instance, channel, sample] {bad[0]}"
        peak = max(peak, float(np.abs(outs[0]).max()))
    assert peak > 0.3
    for b in banks:
        b.close()


@pytest.mark.parametrize("shape", ["16,32", "16,16", "8,32", "32,16"])
def test_staged_pingpong_workgroup_shapes(shape, monkeypatch):
    """The staged kernel for other workgroup shapes (G instances x C samples: KLG_FX_STAGED_G / _C) and without the control path running ahead
    (KLG_FX_STAGED_PIPE=0 for the last): same bits as the hand-written kernel, 130 instances (a last workgroup that is not full), a ragged block."""
    prog, rec = pingpong_program()
    g, c = shape.split(",")
    monkeypatch.setenv("KLG_FX_STAGED", "1"); monkeypatch.setenv("KLG_FX_STAGED_G", g); monkeypatch.setenv("KLG_FX_STAGED_C", c)
    if shape == "32,16":
        monkeypatch.setenv("KLG_FX_STAGED_PIPE", "0")
    K = 130
    staged = klang_amd.FxBank(prog, K, max_block=256, initial_record=rec, channels=2)
    form = staged.graph_form()
    assert form["staged"] and form["instances_per_workgroup"] == int(g) and form["samples_per_chunk"] == int(c), form
    hand = klang_amd.FxBank("pingpong", K, max_block=256)
    rng = np.random.default_rng(5)
    for k in range(0, K, 4):
        for b in (staged, hand):
            b.set_control(k, 5, 0.02 + 0.004 * k); b.set_control(k, 1, 0.02 + 0.004 * k); b.set_control(k, 2, 0.5 * (k % 8 == 0))
    for bi, n in enumerate([256, 256, 200, 256, 256, 256]):
        x = ((rng.random((K, 2, n), dtype=np.float32) - 0.5) * (1.0 if bi < 3 else 0.0)).astype(np.float32)
        a, b = staged.process(x.copy()), hand.process(x.copy())
        assert np.array_equal(bits(a), bits(b)), f"block {bi}: {int((bits(a) != bits(b)).sum())} samples differ"
    staged.close(); hand.close()


def test_staged_reverb_equals_one_lane_per_instance(monkeypatch):
    """Recorded Reverb.k (16 FilteredDelays processed twice per sample, twenty stereo early taps in nested branches, the feedback matrix): the staged
    form against one lane per instance, 70 instances with their own Direct / Early / Mid / Late / Wet dials, dials moved mid-run, block lengths that
    leave ragged chunks."""
    prog = open(os.path.join(GOLDEN, "reverb_recorded.klgg")).read()
    rec = np.array([int(w, 16) for w in open(os.path.join(GOLDEN, "reverb_recorded.rec")).read().split()], np.uint32)
    K = 70
    monkeypatch.setenv("KLG_FX_STAGED", "1")
    staged = klang_amd.FxBank(prog, K, max_block=256, initial_record=rec, channels=2)
    assert staged.graph_form()["staged"], staged.graph_form()["why"]
    monkeypatch.setenv("KLG_FX_STAGED", "0")
    lane = klang_amd.FxBank(prog, K, max_block=256, initial_record=rec, channels=2)
    rng = np.random.default_rng(2)
    for k in range(K):
        for c in range(5):
            v = float(rng.uniform(0.0, 1.0))
            staged.set_control(k, c, v); lane.set_control(k, c, v)
    peak, tail = 0.0, 0.0
    for bi, n in enumerate([256, 48, 256, 17, 256, 256, 100] + [256] * 14):
        if bi == 5:
            for k in range(0, K, 3):
                staged.set_control(k, 2, 0.9); lane.set_control(k, 2, 0.9)
        x = ((rng.random((K, 2, n), dtype=np.float32) - 0.5) * (1.0 if bi < 6 else 0.0)).astype(np.float32)
        a, b = staged.process(x.copy()), lane.process(x.copy())
        bad = np.argwhere(bits(a) != bits(b))
        assert len(bad) == 0, f"block {bi} (n = {n}): {len(bad)} samples differ, first {bad[0]}"
        peak = max(peak, float(np.abs(a).max()))
        if bi >= 12: tail = max(tail, float(np.abs(a).max()))
    assert peak > 1e-2 and tail > 1e-5, f"peak {peak}, tail {tail}: the reflections (50 ms and later) should sound after the input has stopped"
    staged.close(); lane.close()


@pytest.mark.parametrize("name", ["fx_flanger", "fx_topchorus"])
def test_staged_taps_of_a_line_fed_by_in_may_sit_inside_their_own_chunk(name, tmp_path, monkeypatch, capfd):
    """`in >> delay; ... delay(t)` with t swept down to zero (Flanger.k, the shipped Chorus.k): what such a tap reads of its own chunk is the chunk's `in`, taken from
    the chunk's copy in LDS (staged_tap_float_fetch_near) — no ring check, no chunk handed to the plain body.  The staged form against one lane per instance,
    70 instances with their own dials (delays from zero to tens of milliseconds), dials moved mid-run, ragged and one-sample blocks; and once more with the
    path switched off (KLG_FX_STAGED_NEAR=0: the same chunks fail their check and are walked by the plain body)."""
    monkeypatch.setenv("KLANG_MI355_FORCE_GRAPH", "1"); monkeypatch.setenv("KLANG_MI355_DUMP_GRAPH", "1")
    capfd.readouterr()
    run_effect(name, tmp_path)
    err = capfd.readouterr().err
    m = re.search(r"^klgg 1\n.*?^end\n", err, re.S | re.M)
    assert m, "no program in the facade's dump"
    prog = m.group(0)
    r = re.search(r"initial record:((?: [0-9a-f]{8})+)", err)
    rec = np.array([int(w, 16) for w in r.group(1).split()], np.uint32) if r else None
    ch = 2 if "kind effect 2" in prog else 1
    dials = [tuple(float(x) for x in ln.split()[2:5]) for ln in prog.splitlines() if ln.startswith("dial ")]
    K = 70
    banks = []
    for staged, near in (("1", "1"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("KLG_FX_STAGED", staged); monkeypatch.setenv("KLG_FX_STAGED_NEAR", near)
        banks.append(klang_amd.FxBank(prog, K, max_block=256, initial_record=rec, channels=ch))
    assert banks[0].graph_form()["staged"] and banks[1].graph_form()["staged"] and not banks[2].graph_form()["staged"]
    rng = np.random.default_rng(17)
    def move(ks):
        for k in ks:
            for c, (lo, hi, _) in enumerate(dials):
                v = float(lo + (hi - lo) * rng.uniform(0.0, 1.0) ** 2)                  # (squared: many small delays / depths)
                for b in banks: b.set_control(int(k), c, v)
    move(range(K))
    peak = 0.0
    for bi, n in enumerate([256, 37, 1, 100, 256, 33, 256, 250, 7, 256, 256, 31, 256, 256]):
        if bi in (4, 9): move(rng.choice(K, 20, replace=False))
        x = ((rng.random((K, ch, n), dtype=np.float32) - 0.5) * (1.0 if bi < 11 else 0.0)).astype(np.float32)
        outs = [b.process(x.copy()) for b in banks]
        for what, o in zip(("with the chunk's own copy", "with the path switched off"), outs[:2]):
            bad = np.argwhere(bits(o) != bits(outs[2]))
            assert len(bad) == 0, f"block {bi} (n = {n}): staged {what} differs from one lane per instance in {len(bad)} samples, first {bad[0]}"
        peak = max(peak, float(np.abs(outs[0]).max()))
    assert peak > 1e-2
    for b in banks: b.close()


@pytest.mark.parametrize("shape", ["16,16", "16,8"])
def test_staged_reverb_workgroup_shapes(shape, monkeypatch):
    """Recorded Reverb.k with shorter chunks (its 68 values through LDS only fit 16-instance workgroups: every bank size runs it 16 x 32; wider shapes are
    covered by the shipped Chorus.k below): the staged form against one lane per instance, 130 instances (a last workgroup that is not full),
    per-instance dials, a ragged block."""
    prog = open(os.path.join(GOLDEN, "reverb_recorded.klgg")).read()
    rec = np.array([int(w, 16) for w in open(os.path.join(GOLDEN, "reverb_recorded.rec")).read().split()], np.uint32)
    g, c = shape.split(",")
    K = 130
    monkeypatch.setenv("KLG_FX_STAGED", "1"); monkeypatch.setenv("KLG_FX_STAGED_G", g); monkeypatch.setenv("KLG_FX_STAGED_C", c)
    staged = klang_amd.FxBank(prog, K, max_block=256, initial_record=rec, channels=2)
    form = staged.graph_form()
    if not form["staged"]:
        staged.close()
        pytest.skip("this shape does not fit the LDS budget: " + form["why"])
    assert form["instances_per_workgroup"] == int(g) and form["samples_per_chunk"] == int(c), form
    monkeypatch.setenv("KLG_FX_STAGED", "0")
    lane = klang_amd.FxBank(prog, K, max_block=256, initial_record=rec, channels=2)
    rng = np.random.default_rng(23)
    for k in range(K):
        for ctl in range(5):
            v = float(rng.uniform(0.0, 1.0))
            staged.set_control(k, ctl, v); lane.set_control(k, ctl, v)
    peak = 0.0
    for bi, n in enumerate([256, 200, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256]):
        x = ((rng.random((K, 2, n), dtype=np.float32) - 0.5) * (1.0 if bi < 4 else 0.0)).astype(np.float32)
        a, b = staged.process(x.copy()), lane.process(x.copy())
        bad = np.argwhere(bits(a) != bits(b))
        assert len(bad) == 0, f"block {bi} (n = {n}): {len(bad)} samples differ, first {bad[0]}"
        peak = max(peak, float(np.abs(a).max()))
    assert peak > 1e-2
    staged.close(); lane.close()


@pytest.mark.parametrize("shape", ["32,16", "64,8", "16,16"])
def test_staged_chorus_workgroup_shapes(shape, tmp_path, monkeypatch, capfd):
    """The shipped Chorus.k (ten LFO strands: packs of four / two / none by workgroup width; ten taps of lines fed by `in`; a prepare() prologue) in the shapes
    larger banks select: staged against one lane per instance, 130 instances, dials per instance, a ragged block."""
    monkeypatch.setenv("KLANG_MI355_FORCE_GRAPH", "1"); monkeypatch.setenv("KLANG_MI355_DUMP_GRAPH", "1")
    capfd.readouterr()
    run_effect("fx_topchorus", tmp_path)
    err = capfd.readouterr().err
    prog = re.search(r"^klgg 1\n.*?^end\n", err, re.S | re.M).group(0)
    r = re.search(r"initial record:((?: [0-9a-f]{8})+)", err)
    rec = np.array([int(w, 16) for w in r.group(1).split()], np.uint32) if r else None
    dials = [tuple(float(x) for x in ln.split()[2:5]) for ln in prog.splitlines() if ln.startswith("dial ")]
    g, c = shape.split(",")
    K = 130
    monkeypatch.setenv("KLG_FX_STAGED", "1"); monkeypatch.setenv("KLG_FX_STAGED_G", g); monkeypatch.setenv("KLG_FX_STAGED_C", c)
    staged = klang_amd.FxBank(prog, K, max_block=256, initial_record=rec, channels=2)
    form = staged.graph_form()
    assert form["staged"] and form["instances_per_workgroup"] == int(g) and form["samples_per_chunk"] == int(c), form
    monkeypatch.setenv("KLG_FX_STAGED", "0")
    lane = klang_amd.FxBank(prog, K, max_block=256, initial_record=rec, channels=2)
    rng = np.random.default_rng(29)
    for k in range(K):
        for ctl, (lo, hi, _) in enumerate(dials):
            v = float(lo + (hi - lo) * rng.uniform(0.0, 1.0))
            staged.set_control(k, ctl, v); lane.set_control(k, ctl, v)
    peak = 0.0
    for bi, n in enumerate([256, 200, 256, 256, 7, 256, 256]):
        x = ((rng.random((K, 2, n), dtype=np.float32) - 0.5) * (1.0 if bi < 5 else 0.0)).astype(np.float32)
        a, b = staged.process(x.copy()), lane.process(x.copy())
        bad = np.argwhere(bits(a) != bits(b))
        assert len(bad) == 0, f"block {bi} (n = {n}): {len(bad)} samples differ, first {bad[0]}"
        peak = max(peak, float(np.abs(a).max()))
    assert peak > 1e-2
    staged.close(); lane.close()
