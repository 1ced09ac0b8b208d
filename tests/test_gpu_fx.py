"""GPU parity of the effect banks (BASELINE config 4: PingPong.k + Reverb.k) through the C-ABI, against the golden
vectors produced by the genuine reference and against the TEST-ONLY oracle on more instances.
Tolerance: 1e-5 relative to max(|ref|, channel-block peak) (north_star)."""
import glob
import os

import numpy as np
import pytest

from klg_driver import bit_exact_fraction, rel_err, run_fx_scenario_gpu, run_scenario_oracle
from scenario_io import Scenario

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FX_SCENARIOS = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.scn")) if Scenario.load(p).instances > 0 and not os.path.basename(p).startswith("fx_"))   # fx_*: facade-only (test_gpu_fx_facade.py)
TOL = 1e-5


@pytest.mark.parametrize("name", FX_SCENARIOS)
def test_fx_golden(name):
    s = Scenario.load(os.path.join(GOLDEN, name + ".scn"))
    ref = np.load(os.path.join(GOLDEN, name + ".npz"))["per_voice"]
    got = run_fx_scenario_gpu(s)["per_voice"]
    assert got.shape == ref.shape
    err = rel_err(got, ref)
    print(f"{name}: rel err {err:.3e}, bit-exact samples {100 * bit_exact_fraction(got, ref):.2f}%")
    assert err <= TOL


@pytest.mark.parametrize("patch,instances,blocks", [("pingpong", 200, 24), ("reverb", 70, 16)])
def test_fx_against_oracle_many_instances(patch, instances, blocks, oracle_build):
    """More instances than one wave, instance count not a multiple of 64, per-instance controls differ."""
    s = Scenario(patch=patch, block=256, blocks=blocks, instances=instances, burst=3000, seed=77, dump=list(range(0, blocks, 3)))
    rng = np.random.default_rng(3)
    for k in range(instances):
        if patch == "pingpong":
            s.control(0, k, 1, float(rng.uniform(0.01, 0.2)))
            s.control(0, k, 5, float(rng.uniform(0.0, 0.2)))
            s.control(0, k, 0, float(rng.uniform(0.2, 0.9)))
        else:
            s.control(0, k, 2, float(rng.uniform(0.0, 1.0)))
            s.control(0, k, 3, float(rng.uniform(0.0, 1.0)))
            s.control(0, k, 6, float(rng.uniform(0.0, 1.0)))
            s.control(0, k, 7, float(rng.uniform(0.1, 1.0)))
    ref = run_scenario_oracle(s, oracle_build)["per_voice"]
    got = run_fx_scenario_gpu(s)["per_voice"]
    err = rel_err(got, ref)
    exact = bit_exact_fraction(got, ref)
    print(f"{patch}: {instances} instances, rel err {err:.3e}, bit-exact {100 * exact:.2f}%")
    assert err <= TOL
    assert exact == 1.0          # an instance is one lane's (or one quad's) own arithmetic in the reference's order: nothing to round differently


def test_pingpong_near_taps_and_vibrato(oracle_build):
    """Delays shorter than a pipeline chunk (the in-order path of klg_fx_pingpong_x), delays that cross the near/far
    threshold while the smoothed control glides, and LFO vibrato — all against the oracle."""
    K = 130
    s = Scenario(patch="pingpong", block=192, blocks=20, instances=K, burst=2500, seed=11, dump=list(range(0, 20, 2)))
    rng = np.random.default_rng(9)
    for k in range(K):
        s.control(0, k, 0, float(rng.uniform(0.3, 0.95)))
        s.control(0, k, 1, float(rng.choice([0.001, 0.0012, 0.0016, 0.002, 0.004, 0.3])))
        s.control(0, k, 2, float(rng.uniform(0.0, 1.0)))
        s.control(0, k, 3, float(rng.uniform(0.01, 1.0)))
        s.control(0, k, 4, float(rng.uniform(0.0, 1.0)))
    for k in range(0, K, 3):                                   # glide from short to long and back
        s.control(6, k, 1, 0.5)
        s.control(12, k, 1, 0.001)
    ref = run_scenario_oracle(s, oracle_build)["per_voice"]
    got = run_fx_scenario_gpu(s)["per_voice"]
    err = rel_err(got, ref)
    print(f"pingpong near: rel err {err:.3e}, bit-exact {100 * bit_exact_fraction(got, ref):.2f}%")
    assert err <= TOL


@pytest.mark.parametrize("fs,width", [(32000.0, "16"), (32000.0, "32"), (32000.0, "64"), (44100.0, "16"), (96000.0, "16")])
def test_pingpong_near_taps_at_other_sample_rates(fs, width, oracle_build, monkeypatch):
    """How a chunk with a short delay is run depends on the delay IN SAMPLES: at 44.1 / 48 kHz the shortest delay the dial allows (1 ms: the right tap 22 / 24
    samples behind the cursor) leaves every tap further back than half a chunk — two half-chunks, a barrier between them (the "half-far" form); at 32 kHz
    (16 samples) the chunk is walked in sample order by one wave, through LDS for workgroups of 16 / 32 instances, through memory for 64; at 96 kHz 1 ms is
    48 samples: far.  Every form against the oracle, delays that cross the thresholds while the smoothed control glides, vibrato on some."""
    monkeypatch.setenv("KLG_FX_PINGPONG_G", width)
    K = 70
    s = Scenario(patch="pingpong", block=192, blocks=14, instances=K, burst=2500, seed=23, dump=list(range(0, 14, 2)), fs=fs)
    rng = np.random.default_rng(int(fs) + int(width))
    for k in range(K):
        s.control(0, k, 0, float(rng.uniform(0.3, 0.95)))
        s.control(0, k, 1, float(rng.choice([0.001, 0.0011, 0.0013, 0.0016, 0.002, 0.004, 0.3])))
        s.control(0, k, 5, float(rng.choice([0.0, 0.001, 0.0015, 0.003, 0.2])))
        s.control(0, k, 2, float(rng.choice([0.0, 0.0, 0.7])))
        s.control(0, k, 3, float(rng.uniform(0.01, 1.0)))
        s.control(0, k, 4, float(rng.uniform(0.0, 1.0)))
    for k in range(0, K, 3):                                   # glide from short to long and back
        s.control(5, k, 1, 0.4); s.control(5, k, 5, 0.4)
        s.control(9, k, 1, 0.001); s.control(9, k, 5, 0.0)
    ref = run_scenario_oracle(s, oracle_build)["per_voice"]
    got = run_fx_scenario_gpu(s)["per_voice"]
    exact = bit_exact_fraction(got, ref)
    print(f"pingpong near taps at {fs:.0f} Hz, {width} instances per workgroup: bit-exact {100 * exact:.2f}%")
    assert exact == 1.0 and np.abs(ref).max() > 0.05


def test_fx_block_size_independence():
    """Property: 4 x 64-sample blocks == 1 x 256-sample block, bit for bit (PingPong, GPU vs GPU)."""
    def render(block, blocks):
        s = Scenario(patch="pingpong", block=block, blocks=blocks, instances=3, burst=700, seed=5, dump=list(range(blocks)))
        s.ctl = [(0, 0.8), (1, 0.01), (5, 0.01)]
        pv = run_fx_scenario_gpu(s)["per_voice"]           # [B][K][2][N]
        return pv.transpose(1, 2, 0, 3).reshape(3, 2, -1)
    a, b = render(256, 4), render(64, 16)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("block", [16, 32, 48, 64, 96, 128, 176, 256, 512, 1024, 40, 200])
def test_reverb_block_lengths(block, oracle_build):
    """klg_fx_reverb_q picks its way through a block by the block's length and where the rings' cursors stand: blocks of a multiple of sixteen samples
    that start on the grid store their FilteredDelay pieces in PAIRS (128 bytes per line; six-batch turns plus none / two / four batches written out:
    32 -> 2 steady batches, 48 -> 4, 64 -> 6, 96 -> 10, 128 -> 14, 176 -> 20, 256 -> 30, 512 -> 62, 1024 -> 126), a multiple of eight keeps whole 64-byte
    pieces, anything else (40 after the first block, 200) the guarded general form.  Seven instances, ~7,200 samples (early reflections arrive after
    ~2,400, the mid ones ~400 later), a dial changed on the way, against the oracle bit for bit."""
    B = max(2, 7200 // block)
    dump = sorted({0, 1, B // 2, B * 2 // 3, B - 2, B - 1})
    s = Scenario(patch="reverb", block=block, blocks=B, instances=7, burst=2000, seed=31, dump=dump)
    rng = np.random.default_rng(6)
    for k in range(7):
        s.control(0, k, 1, float(rng.uniform(0.3, 1.0))); s.control(0, k, 2, float(rng.uniform(0.2, 1.0))); s.control(0, k, 3, float(rng.uniform(0.2, 1.0)))
        s.control(0, k, 5, float(rng.uniform(2.0, 40.0))); s.control(0, k, 6, float(rng.uniform(0.1, 1.0)))
    s.control(B // 3, 2, 7, 0.4)
    s.sort()
    got = run_fx_scenario_gpu(s)["per_voice"]
    ref = run_scenario_oracle(s, oracle_build)["per_voice"]
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"max abs err {np.abs(got - ref).max()}"
    assert np.abs(got[-1]).max() > 1e-4


def test_reverb_is_independent_of_how_the_stream_is_cut():
    """Property (GPU vs GPU): one stream of 6,144 samples through Reverb.k in blocks of 256 equals the same stream cut into a mixture of lengths —
    which walks klg_fx_reverb_q through all of its ways in and out of a block (pairs, whole pieces, the guarded form) with the rings' cursors on
    and off the store grids — bit for bit, for five instances with different dials."""
    import klang_amd
    K, total = 5, 6144
    rng = np.random.default_rng(12)
    x = rng.uniform(-0.5, 0.5, size=(K, 2, total)).astype(np.float32)
    x[:, :, 2500:] = 0
    dials = [(k, c, float(rng.uniform(lo, hi))) for k in range(K) for c, lo, hi in ((1, 0.3, 1.0), (2, 0.2, 1.0), (3, 0.2, 1.0), (5, 2.0, 40.0), (6, 0.1, 1.0), (7, 0.05, 1.0))]
    def render(cuts):
        bank = klang_amd.FxBank("reverb", K, max_block=1024)
        for k, c, v in dials: bank.set_control(k, c, v)
        out, at = np.zeros_like(x), 0
        for n in cuts:
            io = np.ascontiguousarray(x[:, :, at:at + n]); bank.process(io); out[:, :, at:at + n] = io; at += n
        assert at == total
        bank.close()
        return out
    ref = render([256] * 24)
    mixed = [256, 40, 256, 16, 200, 64, 8, 24, 512, 128, 48, 1024, 96, 32, 176, 1000, 264, 640, 368, 32, 64, 500, 140, 256]
    assert sum(mixed) == total
    got = render(mixed)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"max abs err {np.abs(got - ref).max()}"
    assert np.abs(ref[:, :, 3000:]).max() > 1e-4


def test_fx_silence_in_silence_out():
    import klang_amd
    bank = klang_amd.FxBank("reverb", 5, max_block=128)
    io = np.zeros((5, 2, 128), np.float32)
    for _ in range(3):
        bank.process(io)
    assert not np.any(io)
    bank.close()


@pytest.mark.parametrize("env", [{"KLG_FX_REVERB1": "1"}])
def test_reverb_kernels_agree_bit_for_bit(env, monkeypatch):
    """Two kernels render Reverb.k: the production one (a wave per four instances, a contiguous ring per line) and the single-lane walk of the
    whole graph (KLG_FX_REVERB1=1).  70 instances
    with different controls, a control change mid-run, odd block length: identical bits."""
    def render():
        s = Scenario(patch="reverb", block=200, blocks=20, instances=70, burst=2000, seed=21, dump=list(range(0, 20, 3)))
        rng = np.random.default_rng(8)
        for k in range(70):
            s.control(0, k, 0, float(rng.uniform(0.0, 1.0)))
            s.control(0, k, 2, float(rng.uniform(0.0, 1.0)))
            s.control(0, k, 3, float(rng.uniform(0.0, 1.0)))
            s.control(0, k, 6, float(rng.uniform(0.0, 1.0)))
            s.control(0, k, 7, float(rng.uniform(0.1, 1.0)))
        for k in range(0, 70, 4):
            s.control(9, k, 5, 40.0)
        s.sort()
        return run_fx_scenario_gpu(s)["per_voice"]
    ref = render()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    other = render()
    assert np.abs(ref).max() > 0
    assert np.array_equal(ref.view(np.uint32), other.view(np.uint32))


def test_delay_lines_wrap_around(oracle_build):
    """Run long enough for every ring to wrap (PingPong: 192000 samples; Reverb's FilteredDelay lines advance two positions per sample: 96000
    samples, its early ring 21600): 1024-sample blocks, the oracle walks the same 200k samples."""
    for patch, blocks, dump in (("pingpong", 196, [0, 186, 188, 190, 195]), ("reverb", 100, [0, 21, 22, 93, 94, 95, 99])):
        s = Scenario(patch=patch, block=1024, blocks=blocks, instances=5, burst=400000, seed=3, dump=dump)
        rng = np.random.default_rng(2)
        for k in range(5):
            if patch == "pingpong":
                s.control(0, k, 1, float(rng.uniform(0.05, 0.3))); s.control(0, k, 5, float(rng.uniform(0.05, 0.3))); s.control(0, k, 0, 0.6)
            else:
                s.control(0, k, 2, 0.8); s.control(0, k, 3, 0.7); s.control(0, k, 6, float(rng.uniform(0.2, 1.0)))
        ref = run_scenario_oracle(s, oracle_build)["per_voice"]
        got = run_fx_scenario_gpu(s)["per_voice"]
        err = rel_err(got, ref)
        print(f"{patch} wrap: rel err {err:.3e}, bit-exact {100 * bit_exact_fraction(got, ref):.2f}%")
        assert err <= TOL and np.abs(got).max() > 0


def test_a_control_the_effect_writes_comes_back_and_the_hosts_set_wins():
    """PingPong.k writes controls[1] every sample (`controls[1].set(new_delay)` while controls[5] glides, then the LFO's vibrato on top):
    klg_fx_get_control returns the instance's value after the block — the same from the hand-written kernel and from the RECORDED program
    (tests/golden/pingpong_recorded.klgg: the control is a `ctlvar` word of the record there) — and a klg_fx_set_control overwrites it."""
    import klang_amd
    prog = open(os.path.join(GOLDEN, "pingpong_recorded.klgg")).read()
    rec = np.array([int(w, 16) for w in open(os.path.join(GOLDEN, "pingpong_recorded.rec")).read().split()], np.uint32)
    K, N = 5, 256
    banks = [klang_amd.FxBank("pingpong", K, max_block=N), klang_amd.FxBank(prog, K, max_block=N, initial_record=rec, channels=2)]
    rng = np.random.default_rng(2)
    x = (rng.uniform(-0.5, 0.5, size=(4, K, 2, N))).astype(np.float32)
    seen = []
    for bank in banks:
        for k in range(K):
            bank.set_control(k, 5, 0.1 + 0.15 * k)                     # controls[5] away from `delay`: the scratch branch writes controls[1]
            bank.set_control(k, 2, 0.2 * k); bank.set_control(k, 3, 0.5)   # vibrato: controls[1] += lfo * ...
        vals, outs = [], []
        for b in range(4):
            if b == 2:
                bank.set_control(3, 1, 0.9)                            # the host's set() wins over what the effect wrote
            io = x[b].copy(); bank.process(io); outs.append(io)
            vals.append([bank.get_control(k, 1) for k in range(K)])
        assert bank.get_control(0, 0) == 0.5                           # an ordinary control: the host's value
        seen.append((np.array(vals, np.float32), np.stack(outs)))
        bank.close()
    (va, oa), (vb, ob) = seen
    assert np.array_equal(va.view(np.uint32), vb.view(np.uint32)) and np.array_equal(oa.view(np.uint32), ob.view(np.uint32))
    assert not np.allclose(va[0], 0.5) and len(set(va[-1].tolist())) > 1        # the effect moved them, each instance its own way


@pytest.mark.parametrize("block,width", [(256, 0), (64, 16), (96, 32), (512, 64), (1024, 32), (256, 64), (80, 16)])
def test_pingpong_with_stationary_controls(oracle_build, block, width, monkeypatch):
    """Some ten thousand samples after a dial last moved both control smoothers sit at their fp32 fixed points; klg_fx_pingpong_x then skips the
    serial control chain (every sample's delay time IS the smoothed value), only walks the LFO phase — and, when every tap also lies far enough
    behind the write cursor and the block is whole chunks, runs its audio waves several chunks ahead of themselves (the request-ahead pipeline:
    written out step by step for blocks of 4 / 8 / 16 chunks, a loop otherwise; 80 samples is not whole chunks and stays on the general path).
    40,960 samples with the dials set once (instances differ, some delays too short to qualify), then one dial moved near the end — in and out
    of the stationary state — against the oracle, bit for bit: early blocks (converging), late blocks (stationary), the blocks around the
    change; every workgroup width (KLG_FX_PINGPONG_G)."""
    if width: monkeypatch.setenv("KLG_FX_PINGPONG_G", str(width))
    B = 40960 // block
    change = B * 15 // 16
    dump = sorted({0, 1, B // 4, B * 5 // 8, change - 2, change - 1, change, change + 1, B - 1})
    s = Scenario(patch="pingpong", block=block, blocks=B, instances=20, burst=40960, seed=11, dump=dump)
    rng = np.random.default_rng(8)
    for k in range(20):
        s.control(0, k, 0, float(rng.uniform(0.2, 0.9)))
        s.control(0, k, 1, float(rng.uniform(0.02, 0.6)))
        s.control(0, k, 5, float(rng.uniform(0.02, 0.6)))
        s.control(0, k, 4, float(rng.uniform(0.3, 1.0)))
    s.control(0, 3, 1, 0.003); s.control(0, 3, 5, 0.003)           # 3 ms: taps 72 samples behind the cursor — instance 3's workgroup cannot run ahead, the others do
    for k in range(0, 20, 3):
        s.control(change, k, 5, float(rng.uniform(0.02, 0.6)))
    s.sort()
    got = run_fx_scenario_gpu(s)["per_voice"]
    ref = run_scenario_oracle(s, oracle_build)["per_voice"]
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"max abs err {np.abs(got - ref).max()}"
    assert np.abs(got[-1]).max() > 1e-3


@pytest.mark.parametrize("block,width", [(256, 0), (256, 16), (128, 32), (256, 64), (192, 16), (512, 64), (96, 32)])
def test_pingpong_with_moving_dials(oracle_build, block, width, monkeypatch):
    """Dials under automation: controls[1] / controls[5] of some instance move every few blocks, so the smoothers never settle and every block runs
    klg_fx_pingpong_x's general pipeline (control chain one chunk ahead; whole and ragged last chunks; blocks of 3 .. 16 chunks).  A dial sent to 2 ms
    brings near taps (a chunk walked in order by one wave) for as long as the smoothed delay is that short, a dial-5 move sets off the scratch detector
    (controls[1].set, LFO reset) inside the chain.  70 instances (the second workgroup of every width is partly padding), EVERY block compared with
    the oracle bit for bit.  (Written for a request-ahead variant of this path — control chain three chunks ahead, rows requested two steps early —
    which passed it and was 4 % faster at 4,096 instances, 8 % slower at 16,384: not kept)"""
    if width: monkeypatch.setenv("KLG_FX_PINGPONG_G", str(width))
    B = 15360 // block
    s = Scenario(patch="pingpong", block=block, blocks=B, instances=70, burst=15360, seed=5, dump=list(range(B)))
    rng = np.random.default_rng(21)
    for k in range(70):
        s.control(0, k, 0, float(rng.uniform(0.2, 0.9)))
        s.control(0, k, 1, float(rng.uniform(0.03, 0.6)))
        s.control(0, k, 5, float(rng.uniform(0.03, 0.6)))
        s.control(0, k, 4, float(rng.uniform(0.3, 1.0)))
    for b in range(1, B):
        for _ in range(3):
            s.control(b, int(rng.integers(0, 70)), int(rng.choice([1, 5])), float(rng.uniform(0.03, 0.6)))
    s.control(B // 3, 66, 1, 0.002); s.control(B // 3, 66, 5, 0.002)         # towards 2 ms: near taps in the second workgroup ...
    s.control(2 * B // 3, 66, 1, 0.3); s.control(2 * B // 3, 66, 5, 0.3)     # ... and away again
    s.sort()
    got = run_fx_scenario_gpu(s)["per_voice"]
    ref = run_scenario_oracle(s, oracle_build)["per_voice"]
    bad = [b for b in range(B) if not np.array_equal(got[b].view(np.uint32), ref[b].view(np.uint32))]
    assert not bad, f"blocks {bad[:8]} differ, max abs err {np.abs(got - ref).max()}"
    assert np.abs(got[-1]).max() > 1e-3


@pytest.mark.parametrize("block,width", [(256, 0), (256, 16), (192, 32), (256, 64), (80, 16), (40, 32), (8, 0)])
def test_pingpong_vibrato_with_the_scratch_detector_firing(oracle_build, block, width, monkeypatch):
    """Vibrato on (controls[2], controls[3] > 0): every sample takes the LFO's fp64 sine and writes controls[1].  With fewer than 64 instances per
    workgroup the control wave walks a chunk's LFO phases first, takes the sines several samples at a time in its spare lanes and reads them back in
    the chain; a controls[5] move sets off the scratch detector (`lfo.set(rate, pi)`) for some ten thousand samples, during which the chunk is walked
    the plain way from the first firing sample on (round 3, final form: the phases are walked three chunks ahead and the AUDIO waves take the sines, at
    every width).  Some instances without vibrato in the same workgroup, blocks that are not whole chunks (80, 40) or shorter than the chain's groups
    of eight (8); every block against the oracle, bit for bit."""
    if width: monkeypatch.setenv("KLG_FX_PINGPONG_G", str(width))
    B = 12288 // block
    s = Scenario(patch="pingpong", block=block, blocks=B, instances=40, burst=12288, seed=9, dump=list(range(B)))
    rng = np.random.default_rng(33)
    for k in range(40):
        s.control(0, k, 0, float(rng.uniform(0.2, 0.9)))
        s.control(0, k, 1, float(rng.uniform(0.05, 0.6)))
        s.control(0, k, 2, 0.0 if k % 5 == 4 else float(rng.uniform(0.1, 1.0)))
        s.control(0, k, 3, float(rng.uniform(0.05, 1.0)))
        s.control(0, k, 5, float(rng.uniform(0.05, 0.6)))
    for b in (B // 4, B // 2, B // 2 + 1, 3 * B // 4):
        for k in range(b % 3, 40, 3):
            s.control(b, k, 5, float(rng.uniform(0.05, 0.6)))
    s.sort()
    got = run_fx_scenario_gpu(s)["per_voice"]
    ref = run_scenario_oracle(s, oracle_build)["per_voice"]
    bad = [b for b in range(B) if not np.array_equal(got[b].view(np.uint32), ref[b].view(np.uint32))]
    assert not bad, f"blocks {bad[:8]} differ, max abs err {np.abs(got - ref).max()}"
    assert np.abs(got[-1]).max() > 1e-3


def test_record_download_and_word_upload():
    """klg_fx_download_record / klg_fx_upload_words (what a host-run prepare() uses, include/klang/klang.h EffectBank::host_prepare): a record comes back
    as the device last left it — the dials just set, state a block has changed — and uploaded words are what the next block starts from.  Instances
    3 and 5 get the same dials and the same input and stay bit-equal; a DC-filter state word of 5 is overwritten (the filter sits behind the delay
    lines: their contents stay equal) and its output leaves 3's; instance 3's record uploaded into 5 brings it back, bit for bit."""
    import klang_amd
    K, N = 8, 64
    bank = klang_amd.FxBank("pingpong", K, max_block=N)
    W = bank.record_words()
    rng = np.random.default_rng(4)
    for k in range(K):
        kk = 3 if k == 5 else k
        bank.set_control(k, 0, 0.3 + 0.05 * kk); bank.set_control(k, 1, 0.02 + 0.01 * kk); bank.set_control(k, 4, 0.3)
    r0 = bank.download_record(3)
    assert r0.size == W and np.float32(0.3 + 0.05 * 3) == r0[0:1].view(np.float32)[0] and np.float32(0.02 + 0.01 * 3) == r0[1:2].view(np.float32)[0]   # PingPong: word c is controls[c]
    x = rng.uniform(-0.5, 0.5, size=(K, 2, N)).astype(np.float32)
    x[5] = x[3]
    for _ in range(60): a = bank.process(x.copy())
    assert np.array_equal(a[5].view(np.uint32), a[3].view(np.uint32)) and not np.array_equal(a[3], x[3]) and not np.array_equal(a[4], a[3])
    r3 = bank.download_record(3)
    assert not np.array_equal(r3, r0) and np.array_equal(bank.download_record(5), r3)      # the smoothers, the LFO phase, the DC filters moved — alike in both
    bank.upload_words(5, 11, np.array([0.25], np.float32).view(np.uint32))               # PP_Z: the left DC filter's first state word
    with pytest.raises(klang_amd.KlangError):
        bank.upload_words(5, W - 1, r3[:2])                    # past the end of the record
    b = bank.process(x.copy())
    assert not np.array_equal(b[5, 0], b[3, 0]) and np.array_equal(b[5, 1].view(np.uint32), b[3, 1].view(np.uint32))    # the left channel left, the right did not
    bank.upload_words(5, 0, bank.download_record(3))           # instance 5 becomes instance 3 again
    c = bank.process(x.copy())
    assert np.array_equal(c[5].view(np.uint32), c[3].view(np.uint32))
    assert np.array_equal(bank.download_record(5), bank.download_record(3))
    bank.close()
