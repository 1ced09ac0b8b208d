"""Parity at the BASELINE configs' OWN sizes: cfg 3 = 16,384 SuperSaw voices, cfg 4 = 4,096 PingPong / 4,096 Reverb
instances, the cfg-5 per-GPU share = 131,072 FM4 voices.  The oracle finishes a few thousand voices in seconds, not a hundred thousand,
so each bank is built from CLASSES: synth instance s (effect instance k) plays exactly what instance s mod C (k mod C) plays — same
events, same seeds, same controls, same input.  Then
  * the C class representatives (4,096 voices / 256 instances, spread over the first workgroups) are compared with the oracle,
  * every other instance must equal its representative BIT FOR BIT (they sit in other workgroups, waves and lanes: any
    position-dependent defect of the launch shows up here),
  * the bank's mix must equal the fp64 sum of its own per-voice outputs within the summation-order bound sqrt(V) * eps,
  * and the mix is bit-reproducible from run to run (fixed-order combine, no float atomics)."""
import numpy as np
import pytest

from klg_driver import bit_exact_fraction, rel_err, run_fx_scenario_gpu, run_scenario_gpu, run_scenario_oracle
from scenario_io import EV_CTL, Scenario, fx_input

pytestmark = pytest.mark.gpu
TOL = 1e-5


def class_scenario(patch, classes, notes, blocks, block, seed, seeded, dump):
    """`classes` synth instances, `notes` voices each: note-ons at block 0 (a few later), staggered note-offs (SURVEY §8d's shape, shortened)"""
    s = Scenario(patch=patch, block=block, blocks=blocks, synths=classes, notes=notes, dump=dump)
    rng = np.random.default_rng(seed)
    for sy in range(classes):
        for k in range(notes):
            p = int(rng.integers(36, 97))
            b0 = 0 if k % 5 else 1
            s.on(b0, sy, p, float(rng.uniform(0.25, 1.0)), seed=(1000 * sy + k) if seeded else -1)
            if k % 3 != 2:
                s.off(2 + (k % 4), sy, p)
    s.sort()
    return s


def replicate(s, total_synths):
    """the same events for synth instances c, c + C, c + 2C, ..."""
    C = s.synths
    big = Scenario(patch=s.patch, block=s.block, blocks=s.blocks, synths=total_synths, notes=s.notes, dump=list(s.dump), ctl=list(s.ctl))
    for (b, t, sy, a, bb, seed) in s.ev:
        for r in range(sy, total_synths, C):
            big.ev.append((b, t, r, a, bb, seed))
    big.sort()
    return big


@pytest.mark.parametrize("patch,total_synths,classes,notes,seeded", [("supersaw", 512, 128, 32, True), ("fm4", 4096, 128, 32, False)])
def test_synth_bank_at_config_size(patch, total_synths, classes, notes, seeded, oracle_build):
    dump = [0, 2, 5]
    small = class_scenario(patch, classes, notes, 6, 256, 4242, seeded, dump)
    ref = run_scenario_oracle(small, oracle_build)
    big = replicate(small, total_synths)
    got = run_scenario_gpu(big)
    V, Vc = big.voices, small.voices
    pv = got["per_voice"]                                            # [dumps][V][N]
    assert pv.shape[1] == V
    # (1) class representatives against the oracle
    err = rel_err(pv[:, :Vc], ref["per_voice"])
    exact = bit_exact_fraction(pv[:, :Vc], ref["per_voice"])
    print(f"{patch}: {V} voices; {Vc} class voices vs oracle: rel err {err:.3e}, bit-exact {100 * exact:.3f} %")
    assert err <= TOL
    assert np.array_equal(got["stages"][:, :Vc], ref["stages"])
    # (2) every replica equals its representative bit for bit
    reps = pv.reshape(len(dump), total_synths // classes, Vc, -1)
    assert np.array_equal(reps.view(np.uint32), np.broadcast_to(reps[:, :1], reps.shape).view(np.uint32)), "a replica differs from its class representative"
    st = got["stages"].reshape(big.blocks, total_synths // classes, Vc)
    assert np.array_equal(st, np.broadcast_to(st[:, :1], st.shape))
    # (3) the mix is the sum of the voices (fp64 reference sum of the bank's own per-voice outputs)
    for i, b in enumerate(dump):
        want = pv[i].astype(np.float64).sum(axis=0)
        peak = float(np.abs(pv[i]).max())
        bound = 4 * np.sqrt(V) * np.finfo(np.float32).eps * peak * np.sqrt(V)      # partial sums grow like sqrt(V) * peak; each add rounds at eps of that
        assert float(np.abs(got["mix"][b, 0] - want).max()) <= bound, f"block {b}: {np.abs(got['mix'][b, 0] - want).max()} > {bound}"
    assert float(np.abs(got["mix"]).max()) > 0


def test_mix_is_bit_reproducible_from_run_to_run():
    """16,384 sub2a voices, every wave of every workgroup live: two fresh banks produce the same mix BIT FOR BIT (the four waves of a
    workgroup add their chunk sums into rows of their own, combined in wave order — no float atomics anywhere)."""
    small = class_scenario("sub2a", 8, 128, 3, 256, 7, False, [])
    big = replicate(small, 128)
    a = run_scenario_gpu(big, per_voice=False)["mix"]
    b = run_scenario_gpu(big, per_voice=False)["mix"]
    assert np.abs(a).max() > 0
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def fx_class_run(patch, K, C, blocks, block, oracle_build, dump, vibrato=True):
    s = Scenario(patch=patch, block=block, blocks=blocks, instances=C, burst=3 * block, seed=99, dump=dump)
    rng = np.random.default_rng(5)
    for k in range(C):
        if patch == "pingpong":
            s.control(0, k, 1, float(rng.uniform(0.02, 0.3)))
            s.control(0, k, 5, float(rng.uniform(0.02, 0.3)))
            s.control(0, k, 0, float(rng.uniform(0.2, 0.9)))
            if vibrato and k % 7 == 0:
                s.control(0, k, 2, 0.4); s.control(0, k, 3, 0.3)           # some vibrato
        else:
            s.control(0, k, 2, float(rng.uniform(0.0, 1.0)))
            s.control(0, k, 3, float(rng.uniform(0.0, 1.0)))
            s.control(0, k, 6, float(rng.uniform(0.0, 1.0)))
            s.control(0, k, 7, float(rng.uniform(0.1, 1.0)))
    for k in range(0, C, 9):
        s.control(3, k, 6 if patch == "reverb" else 0, 0.5)                # a control change mid-run
    s.sort()
    ref = run_scenario_oracle(s, oracle_build)["per_voice"]               # [dumps][C][2][N]
    import klang_amd
    bank = klang_amd.FxBank(patch, K, fs=s.fs, max_block=block)
    got = []
    evi = 0
    t = np.arange(block, dtype=np.uint64)
    try:
        for b in range(blocks):
            while evi < len(s.ev) and s.ev[evi][0] <= b:
                _, ty, inst, a, bb, _seed = s.ev[evi]
                if ty == EV_CTL:
                    for r in range(inst, K, C):
                        bank.set_control(r, int(a), bb)
                evi += 1
            cls = np.empty((C, 2, block), np.float32)
            for k in range(C):
                for ch in range(2):
                    cls[k, ch] = fx_input(s.seed, k, ch, t + np.uint64(b * block), s.burst)
            io = np.ascontiguousarray(np.tile(cls, (K // C, 1, 1)))
            bank.process(io)
            if b in dump:
                got.append(io.copy())
    finally:
        bank.close()
    return np.stack(got), ref


@pytest.mark.parametrize("patch,blocks,vibrato", [("pingpong", 8, True), ("pingpong", 96, False), ("reverb", 16, True)])          # (the first early reflection of Reverb.k arrives after 50 ms = 9 blocks)
def test_effect_bank_at_config_size(patch, blocks, vibrato, oracle_build):
    """cfg 4 at its own size: 4,096 instances (Reverb: 51 GB of delay lines), 256 classes against the oracle, 3,840 replicas bit for bit.
    The 96-block PingPong run has no vibrato anywhere: after some sixty-five blocks both control smoothers of every instance stand at their fixed
    points and every workgroup runs the request-ahead pipeline (klg_fx_pingpong_x, stationary blocks) — the blocks compared are the first, two in
    the converging phase, and five of the last thirty."""
    K, C, N = 4096, 256, 256
    dump = [0, blocks // 2 + 2, blocks - 1] if blocks < 40 else [0, 20, 50, 66, 75, 80, 90, blocks - 1]
    got, ref = fx_class_run(patch, K, C, blocks, N, oracle_build, dump, vibrato)
    err = rel_err(got[:, :C], ref)
    print(f"{patch}: {K} instances; {C} class instances vs oracle: rel err {err:.3e}, bit-exact {100 * bit_exact_fraction(got[:, :C], ref):.3f} %")
    assert err <= TOL
    reps = got.reshape(len(dump), K // C, C, 2, N)
    assert np.array_equal(reps.view(np.uint32), np.broadcast_to(reps[:, :1], reps.shape).view(np.uint32)), "a replica differs from its class representative"
    assert np.abs(got).max() > 0


def test_recorded_pingpong_equals_its_kernel_at_config_size():
    """Config 4's size, 4,096 instances: the shipped examples/PingPong.k RECORDED (tests/golden/pingpong_recorded.klgg + .rec, what the facade
    records from the unchanged file) against its hand-written kernel klg_fx_pingpong_x on the same input, dials spread over their ranges on every
    seventh instance (vibrato on, taps inside a chunk, the scratch branch writing controls[1]) — bit for bit, and the control the effect writes
    comes back the same from both."""
    import os
    import klang_amd
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    prog = open(os.path.join(golden, "pingpong_recorded.klgg")).read()
    rec = np.array([int(w, 16) for w in open(os.path.join(golden, "pingpong_recorded.rec")).read().split()], np.uint32)
    K, N, B = 4096, 256, 6
    rng = np.random.default_rng(41)
    ctl = [(k, c, float(rng.uniform(lo, hi))) for k in range(0, K, 7) for c, lo, hi in ((0, 0.2, 0.9), (1, 0.002, 0.6), (2, 0.0, 1.0), (3, 0.01, 1.0), (4, 0.2, 1.5), (5, 0.0, 0.4))]
    x = rng.uniform(-0.5, 0.5, size=(B, K, 2, N)).astype(np.float32)
    x[4:] = 0
    outs, written = [], []
    for bank in (klang_amd.FxBank("pingpong", K, max_block=N), klang_amd.FxBank(prog, K, max_block=N, initial_record=rec, channels=2)):
        for k, c, v in ctl:
            bank.set_control(k, c, v)
        res = []
        for b in range(B):
            io = x[b].copy()
            bank.process(io)
            res.append(io)
        outs.append(np.stack(res))
        written.append(np.array([bank.get_control(k, 1) for k in range(0, K, 97)], np.float32))
        bank.close()
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)), f"max abs diff {np.abs(outs[0] - outs[1]).max()}"
    assert np.array_equal(written[0].view(np.uint32), written[1].view(np.uint32))
    assert np.isfinite(outs[0]).all() and np.abs(outs[0][5]).max() > 1e-3                # the echoes are still sounding after the input stopped
