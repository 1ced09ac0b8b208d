"""klg_fx_render_device: a span of blocks of an effect bank in one call — the host's block loop around Stereo::Effect::process(Stereo::buffer)
(templates/juce/effect/Source/PluginProcessor.cpp:153-178 called once per block) for a stream known in advance.  Whatever the library does with the span
(PingPong's pipeline running across the block boundaries in ONE launch, the staged form of a recorded effect walking the blocks itself with prepare() at the
head of each, Reverb's launches back to back) the samples and the state it leaves are those of block-by-block calls, bit for bit."""
import os

import numpy as np
import pytest
import torch

import klang_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# an effect of our own whose prepare() is NOT idempotent: `tape.set(controls[0] * 2400)` places the read head once per block (the per-block prologue),
# process() is `in >> tape; out = in + tape` — Delay::process walks the head.  A span must re-place the head at every block boundary.
TAPE = """klgg 1
kind effect 1
ctl 1
dial 0 0 1 0.5
node 0 delay 4800
op ctl 0 -1 -1 -1 00000000
op const 1 -1 -1 -1 45160000
op mul 2 0 1 -1 00000000
op delayset -1 2 -1 0 00000000
op in 3 -1 -1 -1 00000000
op delayin -1 3 -1 0 00000000
op delayout 4 -1 -1 0 00000000
op add 5 3 4 -1 00000000
prepare 4
ret 5
end
"""


def recorded(name):
    prog = open(os.path.join(GOLDEN, name + ".klgg")).read()
    rec = np.array([int(w, 16) for w in open(os.path.join(GOLDEN, name + ".rec")).read().split()], np.uint32)
    return prog, rec


def make(kind, K, n):
    if kind == "pingpong":
        return klang_amd.FxBank("pingpong", K, max_block=n), 2
    if kind == "reverb":
        return klang_amd.FxBank("reverb", K, max_block=n), 2
    if kind == "pingpong_recorded":
        prog, rec = recorded("pingpong_recorded")
        return klang_amd.FxBank(prog, K, max_block=n, initial_record=rec, channels=2), 2
    if kind == "tape":
        return klang_amd.FxBank(TAPE, K, max_block=n, channels=1), 1
    raise ValueError(kind)


@pytest.mark.parametrize("kind,n,spans", [("pingpong", 256, (1, 5, 3, 8)), ("pingpong", 96, (4, 7)), ("pingpong", 100, (3, 3)), ("pingpong", 512, (2, 3)),
                                          ("pingpong_recorded", 256, (4, 3)), ("pingpong_recorded", 80, (5, 2)), ("reverb", 256, (3, 2)), ("tape", 64, (6, 3)), ("tape", 50, (4, 2))])
def test_a_span_of_blocks_equals_block_by_block(kind, n, spans):
    K = 70
    rng = np.random.default_rng(7)
    a, CH = make(kind, K, n)
    b, _ = make(kind, K, n)
    if kind.startswith("pingpong"):
        for k in range(0, K, 3):
            for bank in (a, b):
                bank.set_control(k, 5, 0.01 + 0.005 * k); bank.set_control(k, 1, 0.01 + 0.005 * k); bank.set_control(k, 0, 0.3 + 0.008 * k)
                if k % 9 == 0: bank.set_control(k, 2, 0.6)
    elif kind == "reverb":                                                       # (Direct up: the first reflection is 50 ms = ten blocks away)
        for k in range(K):
            for bank in (a, b): bank.set_control(k, 0, 0.3 + 0.01 * k); bank.set_control(k, 2, 0.5); bank.set_control(k, 3, 0.4)
    elif kind == "tape":
        for k in range(K):
            for bank in (a, b): bank.set_control(k, 0, 0.02 + 0.013 * k)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for si, blocks in enumerate(spans):
            x = (torch.from_numpy(rng.random((blocks, K, CH, n), dtype=np.float32)) - 0.5).cuda() * (1.0 if si < len(spans) - 1 else 0.25)
            ya, yb = x.clone(), x.clone()
            a.render_device(ya.data_ptr(), blocks, n, st.cuda_stream)
            for blk in range(blocks):
                b.process_device(yb[blk].data_ptr(), n, st.cuda_stream)
            st.synchronize()
            bad = (ya.view(torch.int32) != yb.view(torch.int32)).nonzero()
            assert len(bad) == 0, f"span {si} ({blocks} blocks of {n}): {len(bad)} samples differ, first [block, instance, channel, sample] {bad[0].tolist()}"
            assert float(ya.abs().max()) > 1e-3
            if kind.startswith("pingpong") and si == 0:                           # a dial moved between two spans
                for bank in (a, b): bank.set_control(1, 5, 0.31); bank.set_control(1, 1, 0.31)
    if kind in ("pingpong_recorded", "tape"):                                    # ... and the same state afterwards
        for k in (0, 1, K - 1):
            assert np.array_equal(a.download_record(k), b.download_record(k))
    a.close(); b.close()


@pytest.mark.parametrize("width", [16, 32, 64])
def test_spans_with_moving_dials_equal_block_by_block(width, monkeypatch):
    """Spans whose dials are NOT at rest (no vibrato anywhere): klg_fx_pingpong_x runs the control chain P + 1 chunks ahead of the audio and the
    request-ahead pipeline reads every sample's delay time from its buffers (klg_fx.hpp, `moving`; spans of >= PPX_MOVING_MIN = 32 chunks, widths 16 / 32).
    Between spans: dials turned (the scratch detector fires, controls[1] is re-set, both smoothers converge again), one instance taken to a delay too near
    to run ahead (its workgroup falls back to the general loop), one to the longest delay the dial allows.  Against the same blocks one by one (the
    general loop, which the oracle tests cover), bit for bit, every span; and the instances' records afterwards.  Later spans: controls[5] at rest in every instance
    (the first half of the chain is then not run), only the time dial turned."""
    monkeypatch.setenv("KLG_FX_PINGPONG_G", str(width))
    K, n = 70, 256
    spans = (6, 4, 12, 3, 9, 20, 20, 20, 10, 7)                                  # (the 20s: controls[5] comes to rest everywhere; then the time dial alone is turned)
    rng = np.random.default_rng(31)
    a, CH = make("pingpong", K, n)
    b, _ = make("pingpong", K, n)
    def dial(k, idx, v):
        for bank in (a, b): bank.set_control(k, idx, v)
    for k in range(K):
        t = float(rng.uniform(0.03, 0.7))
        dial(k, 5, t); dial(k, 1, t); dial(k, 0, float(rng.uniform(0.2, 0.9))); dial(k, 4, float(rng.uniform(0.3, 1.0)))
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for si, blocks in enumerate(spans):
            x = (torch.from_numpy(rng.random((blocks, K, CH, n), dtype=np.float32)) - 0.5).cuda()
            ya, yb = x.clone(), x.clone()
            a.render_device(ya.data_ptr(), blocks, n, st.cuda_stream)
            for blk in range(blocks):
                b.process_device(yb[blk].data_ptr(), n, st.cuda_stream)
            st.synchronize()
            bad = (ya.view(torch.int32) != yb.view(torch.int32)).nonzero()
            assert len(bad) == 0, f"span {si} ({blocks} blocks of {n}): {len(bad)} samples differ, first [block, instance, channel, sample] {bad[0].tolist()}"
            assert float(ya.abs().max()) > 1e-3
            if si == 0:
                for k in range(0, K, 5): dial(k, 5, float(rng.uniform(0.03, 0.9)))      # scratches
                dial(7, 1, 0.5)                                                        # the time dial alone
            elif si == 1:
                dial(33, 5, 0.002)                                                     # heading to a near delay: this workgroup may not run ahead
                dial(50, 5, 1.0); dial(2, 5, 0.011)                                    # the longest; one whose taps get close
            elif si == 2:
                dial(33, 5, 0.4); dial(12, 4, 0.1); dial(13, 0, 0.95)
            elif si == 3:
                for k in range(1, K, 7): dial(k, 5, float(rng.uniform(0.05, 0.3)))
            elif si == 7:
                for k in range(2, K, 6): dial(k, 1, float(rng.uniform(0.05, 0.9)))     # controls[5] at rest, controls[1].smooth() on its way: the first control wave has nothing to do
            elif si == 8:
                dial(4, 1, 0.004); dial(60, 1, 1.0)
    for k in (0, 7, 33, 50, K - 1):
        assert np.array_equal(a.download_record(k), b.download_record(k))
    a.close(); b.close()


@pytest.mark.parametrize("kind,width", [("pingpong", 16), ("pingpong", 32), ("pingpong", 64), ("pingpong_recorded", 0)])
def test_stationary_spans_after_convergence(kind, width, oracle_build, monkeypatch):
    """The path the cfg-4 bench legs time: a span on a bank whose dials have been still for so long that both Control::smooth chains sit at their fp32 fixed
    points (klang.h:1708, 1715: some ten thousand samples from a fresh object) — klg_fx_pingpong_x then takes the span as ONE launch whose request-ahead
    pipeline runs across the block boundaries, in groups of steps without run-time guards (klg_fx.hpp; for the recorded PingPong.k: the staged kernel walking
    the blocks).  Two banks, 80 blocks of 256 block by block on both; then spans of 24 and 64 blocks on one, the same blocks one by one on the other: bit for
    bit, every block; and blocks of both spans against the oracle.  70 instances: in one workgroup an instance at 3 ms (taps too near to run ahead), in another
    one at 1 ms — the shortest the dial allows: half-far chunks — and 1.5 / 2 ms; the other workgroups run ahead.  Every workgroup width."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from klg_driver import run_scenario_oracle
    from scenario_io import Scenario, fx_input
    if width: monkeypatch.setenv("KLG_FX_PINGPONG_G", str(width))
    K, n, warm, spans = 70, 256, 80, (24, 64)
    B = warm + sum(spans)
    dump = [0, warm - 1, warm, warm + 5, warm + spans[0] - 1, warm + spans[0], warm + spans[0] + 40, B - 1]
    s = Scenario(patch="pingpong", block=n, blocks=B, instances=K, burst=B * n, seed=23, dump=dump)
    rng = np.random.default_rng(17)
    near = {3: 0.003, 17: 0.001, 40: 0.0015, 41: 0.002}
    for k in range(K):
        s.control(0, k, 0, float(rng.uniform(0.2, 0.9)))
        t = near.get(k, float(rng.uniform(0.02, 0.6)))
        s.control(0, k, 1, t if k in near else float(rng.uniform(0.02, 0.6)))
        s.control(0, k, 5, t)
        s.control(0, k, 4, float(rng.uniform(0.3, 1.0)))
    s.sort()
    a, CH = make(kind, K, n)
    b, _ = make(kind, K, n)
    for bank in (a, b):
        for (_blk, _ty, inst, idx, val, _seed) in s.ev:
            bank.set_control(inst, int(idx), val)
    t = np.arange(B * n, dtype=np.uint64)
    x = np.empty((K, 2, B * n), np.float32)
    for k in range(K):
        for ch in range(2):
            x[k, ch] = fx_input(s.seed, k, ch, t, s.burst)
    x = np.ascontiguousarray(x.reshape(K, 2, B, n).transpose(2, 0, 1, 3))          # [block][instance][channel][n]
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ya, yb = torch.from_numpy(x).cuda(), torch.from_numpy(x).cuda()
        for blk in range(warm):
            a.process_device(ya[blk].data_ptr(), n, st.cuda_stream); b.process_device(yb[blk].data_ptr(), n, st.cuda_stream)
        at = warm
        for blocks in spans:
            a.render_device(ya[at].data_ptr(), blocks, n, st.cuda_stream)
            for blk in range(at, at + blocks):
                b.process_device(yb[blk].data_ptr(), n, st.cuda_stream)
            at += blocks
        st.synchronize()
    bad = (ya.view(torch.int32) != yb.view(torch.int32)).nonzero()
    assert len(bad) == 0, f"{len(bad)} samples differ between spans and single blocks, first [block, instance, channel, sample] {bad[0].tolist()}"
    ref = run_scenario_oracle(s, oracle_build)["per_voice"]                         # [dumped block][instance][channel][n]
    got = ya.cpu().numpy()[dump]
    wrong = [dump[i] for i in range(len(dump)) if not np.array_equal(got[i].view(np.uint32), ref[i].view(np.uint32))]
    assert not wrong, f"blocks {wrong} differ from the oracle, max abs err {np.abs(got - ref).max()}"
    assert np.abs(got[-1]).max() > 1e-3
    a.close(); b.close()
