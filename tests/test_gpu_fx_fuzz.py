"""tests/test_gpu_fx_fuzz.py — random effect programs (tests/fx_fuzz.py) through BOTH generated kernels of the same program: klg_fx_staged (the sample-parallel
form: levels, the control path a chunk ahead, ring checks, the parts of a failed chunk, taps served from the chunk's own copy of `in`) against
klg_fx_graph<P> (one lane per instance, the samples in order: the form that is pinned to the genuine header by every example effect's fixture) — bit for bit,
every instance, every sample, over ragged blocks with dials that move between blocks and delay times swept through the inside of a chunk.
The staged form is also run with its retry switched off and in a second workgroup shape for a third of the programs."""
import os

import numpy as np
import pytest

import klang_amd
from fx_fuzz import program

pytestmark = pytest.mark.gpu

BLOCKS = [256, 37, 1, 100, 32, 33, 64, 250, 7, 256, 16, 8, 9, 256, 96, 256]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("seed", list(range(48)))
def test_random_effect_program_staged_equals_one_lane_per_instance(seed, monkeypatch):
    CH = 1 if seed % 5 == 4 else 2
    prog, what, dials = program(seed, channels=CH)
    K = 40
    forms = [("staged", {"KLG_FX_STAGED": "1"}), ("one lane per instance", {"KLG_FX_STAGED": "0"})]
    if seed % 3 == 0:
        forms.append(("staged, a failed chunk straight to the plain body", {"KLG_FX_STAGED": "1", "KLG_FX_STAGED_RETRY": "0"}))
    if seed % 3 == 1:
        forms.append(("staged, 32 instances x 16 samples", {"KLG_FX_STAGED": "1", "KLG_FX_STAGED_G": "32", "KLG_FX_STAGED_C": "16"}))
    banks = []
    for name, env in forms:
        for k in ("KLG_FX_STAGED", "KLG_FX_STAGED_RETRY", "KLG_FX_STAGED_G", "KLG_FX_STAGED_C"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        banks.append(klang_amd.FxBank(prog, K, max_block=256, channels=CH))
    f = banks[0].graph_form()
    assert f["staged"], f"seed {seed} ({what}): no sample-parallel form: {f['why']}"
    assert not banks[1].graph_form()["staged"]
    rng = np.random.default_rng(1000 + seed)

    def dial(k, c, v):
        for b in banks:
            b.set_control(k, c, float(v))
    lo, hi = dials[1][0], dials[1][1]
    for k in range(K):
        dial(k, 0, rng.uniform(0.1, 0.9))
        dial(k, 1, [rng.uniform(lo, hi), rng.uniform(1.0, 40.0), rng.uniform(6.0, 24.0), float(rng.integers(1, 34))][k % 4])   # a chunk is 32 samples: times around and inside it
        dial(k, 2, rng.uniform(0.0, 12.0) if k % 2 else 0.0)
        dial(k, 3, rng.uniform(0.0, 1.0))
    peak = 0.0
    for bi, n in enumerate(BLOCKS):
        if bi in (3, 7, 10, 13):
            for k in rng.choice(K, 15, replace=False):
                dial(int(k), 1, rng.uniform(1.0, min(hi, 60.0)))
                dial(int(k), 2, rng.uniform(0.0, 12.0))
        x = ((rng.random((K, CH, n), dtype=np.float32) - 0.5) * (1.0 if bi < 13 else 0.0)).astype(np.float32)
        outs = [b.process(x.copy()) for b in banks]
        assert np.isfinite(outs[1]).all(), f"seed {seed} ({what}): the program blew up in block {bi}"
        for (name, _), o in zip(forms, outs):
            bad = np.argwhere(bits(o) != bits(outs[1]))
            assert len(bad) == 0, (f"seed {seed} ({what}), block {bi} (n = {n}): '{name}' differs from one lane per instance in {len(bad)} samples, "
                                   f"first [instance, channel, sample] {bad[0]}: {o[tuple(bad[0])]!r} against {outs[1][tuple(bad[0])]!r}")
        peak = max(peak, float(np.abs(outs[1]).max()))
    assert peak > 1e-2
    for c in range(4):
        for k in (0, 1, K - 1):
            assert banks[0].get_control(k, c) == banks[1].get_control(k, c)
    for b in banks:
        b.close()


def test_every_way_through_the_parts_of_a_failed_chunk_is_taken(monkeypatch, capfd):
    """Three of the random programs (feedback / head lines: taps checked against the chunk's own rows), one workgroup each (16 instances) with delay times
    that wobble through the inside of a chunk (the dial at 8 .. 40 samples, the LFO's depth at 10), the kernel's own counters on (KLG_FX_STAGED_STAMP=1: workgroup 0
    prints them after every launch): chunks that failed their ring check, parts that passed, parts cut in two, parts walked by the plain body, and catch-ups
    of the control path in front of a plain walk inside a chunk (a part passed, a later one of the same chunk did not) — every one of them must have
    happened, with the output equal to the one-lane kernel's, bit for bit."""
    import re
    total = np.zeros(5, np.int64)
    launches = 0
    for seed in (0, 6, 20):
        prog, what, dials = program(seed)
        K = 16
        monkeypatch.setenv("KLG_FX_STAGED", "1"); monkeypatch.setenv("KLG_FX_STAGED_STAMP", "1")
        staged = klang_amd.FxBank(prog, K, max_block=256, channels=2)
        assert staged.graph_form()["staged"]
        monkeypatch.delenv("KLG_FX_STAGED_STAMP"); monkeypatch.setenv("KLG_FX_STAGED", "0")
        lane = klang_amd.FxBank(prog, K, max_block=256, channels=2)
        rng = np.random.default_rng(50 + seed)
        def dial(k, c, v):
            staged.set_control(k, c, float(v)); lane.set_control(k, c, float(v))
        # (a chunk passes or fails for the workgroup as a whole: the sixteen instances move together — one delay time, one LFO rate — so that what the
        #  nearest tap of the workgroup does changes from part to part)
        for k in range(K):
            dial(k, 0, rng.uniform(0.3, 0.9)); dial(k, 1, 30.0); dial(k, 2, 10.0); dial(k, 3, 0.5)
        capfd.readouterr()
        for bi in range(48):
            if bi % 3 == 2:                                                       # down through the chunk's length, and up again: the smoothed time sweeps for ~4 blocks
                t = rng.uniform(4.0, 14.0) if (bi // 3) % 2 == 0 else rng.uniform(25.0, 40.0)
                for k in range(K):
                    dial(k, 1, t)
            x = (rng.random((K, 2, 256), dtype=np.float32) - 0.5).astype(np.float32)
            a, b = staged.process(x.copy()), lane.process(x.copy())
            bad = np.argwhere(bits(a) != bits(b))
            assert len(bad) == 0, f"seed {seed} ({what}), block {bi}: the staged form differs from one lane per instance in {len(bad)} samples, first {bad[0]}"
        staged.sync(); staged.close(); lane.close()
        rows = re.findall(r"staged parts: (\d+) chunks failed their check, (\d+) parts passed, (\d+) cut in two, (\d+) walked by the plain body, (\d+) control catch-ups", capfd.readouterr().out)
        assert len(rows) >= 24, f"seed {seed}: the kernel's counters were not printed"
        launches += len(rows)
        total += np.array(rows, dtype=np.int64).sum(axis=0)
    with capfd.disabled():
        print(f"parts of failed chunks over {launches} launches: failed chunks {total[0]}, parts passed {total[1]}, cut in two {total[2]}, plain walks {total[3]}, control catch-ups {total[4]}")
    assert (total[:4] > 50).all() and total[4] >= 1, total


@pytest.mark.parametrize("seed,n", [(2, 96), (6, 256), (10, 80), (16, 128), (18, 33), (21, 256)])
def test_random_effect_program_a_span_in_one_launch_equals_one_lane_block_by_block(seed, n, monkeypatch):
    """klg_fx_render_device on the staged form — ONE launch walks the span's blocks, prepare() at the head of each (a read head placed in prepare() is placed
    again), the parts of failed chunks inside it — against the one-lane kernel fed the same blocks one call at a time; dials moved between the spans;
    the records both leave are the same words."""
    import torch
    prog, what, dials = program(seed)
    K = 40
    monkeypatch.setenv("KLG_FX_STAGED", "1")
    staged = klang_amd.FxBank(prog, K, max_block=n, channels=2)
    assert staged.graph_form()["staged"]
    monkeypatch.setenv("KLG_FX_STAGED", "0")
    lane = klang_amd.FxBank(prog, K, max_block=n, channels=2)
    rng = np.random.default_rng(2000 + seed)
    def dial(k, c, v):
        staged.set_control(k, c, float(v)); lane.set_control(k, c, float(v))
    for k in range(K):
        dial(k, 0, rng.uniform(0.1, 0.9)); dial(k, 1, rng.uniform(1.0, 40.0)); dial(k, 2, rng.uniform(0.0, 12.0) if k % 2 else 0.0); dial(k, 3, rng.uniform(0.0, 1.0))
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for si, blocks in enumerate((5, 1, 7, 3)):
            x = (torch.from_numpy(rng.random((blocks, K, 2, n), dtype=np.float32)) - 0.5).cuda()
            ya, yb = x.clone(), x.clone()
            staged.render_device(ya.data_ptr(), blocks, n, st.cuda_stream)
            for blk in range(blocks):
                lane.process_device(yb[blk].data_ptr(), n, st.cuda_stream)
            st.synchronize()
            bad = (ya.view(torch.int32) != yb.view(torch.int32)).nonzero()
            assert len(bad) == 0, f"seed {seed} ({what}), span {si} ({blocks} blocks of {n}): {len(bad)} samples differ, first [block, instance, channel, sample] {bad[0].tolist()}"
            for k in rng.choice(K, 10, replace=False):
                dial(int(k), 1, rng.uniform(1.0, 40.0))
    for k in (0, 1, K - 1):
        assert np.array_equal(staged.download_record(k), lane.download_record(k)), f"seed {seed} ({what}): instance {k}'s record"
    staged.close(); lane.close()
