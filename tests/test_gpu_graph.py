"""Graph patches (include/klang_mi355_graph.h): a recorded process() body compiled with hipRTC must render
bit-for-bit what the hand-written kernel of the same patch renders (library level; the DSL facade is test_gpu_facade.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SUB2A_PROGRAM = """klgg 1
ctl 0
node 0 saw
node 1 lpf
node 2 adsr
op osc 0 -1 -1 0 0        # osc >> lpf >> out
op lpf 1 0 -1 1 0
op env 2 -1 -1 2 0        # out *= adsr++
op mul 3 1 2 -1 0
op stopif -1 -1 -1 2 0    # if (adsr.finished()) stop();
ret 3
end
"""


def sub2a_to_graph(r):
    """rec::Sub2a (20 words: flags | OsmRec | BiquadRec | AdsrRec) -> the graph record of SUB2A_PROGRAM (25 words)."""
    flags = int(r[0])
    g = np.zeros(25, np.uint32)
    g[0] = flags & 3
    g[1:5] = r[1:5]; g[5] = (flags >> 8) & 3                       # osm: inc offset duty delta | state | frequency cache (unused: no set(f) in process())
    g[7:14] = r[5:12]                                              # lpf: b0 b1 b2 a1 a2 z0 z1 (f, Q unused)
    g[16:20] = r[12:16]; g[20] = (flags >> 2) & 0x3F; g[21:25] = r[16:20]   # adsr: ramp, time | bits | A AD S R
    return g


@pytest.mark.parametrize("lanes", ["two voices per lane", "one voice per lane", "a voice per wave", "eight voices per wave"])
def test_graph_sub2a_equals_handwritten_kernel(lanes, monkeypatch):
    """The generated patch runs two voices per lane (packed primitives, klg_render_x2<P>) when every node has a packed form —
    this program does — and one per lane with KLG_GRAPH_X1=1; a bank as small as this one takes the sample-parallel form by default
    (klg_render_gsp<PatchGen>, klg_render_sp.hpp; KLG_GRAPH_SP = 0 / 1 / 8).  All must equal the hand-written kernel bit for bit."""
    import klang_amd
    monkeypatch.setenv("KLG_GRAPH_SP", {"a voice per wave": "1", "eight voices per wave": "8"}.get(lanes, "0"))
    if lanes == "one voice per lane":
        monkeypatch.setenv("KLG_GRAPH_X1", "1")
    S, P, N = 3, 32, 192
    hand = klang_amd.SynthBank("sub2a", synths=S, notes=P, max_block=N)
    gen = klang_amd.SynthBank(SUB2A_PROGRAM, synths=S, notes=P, max_block=N)
    assert gen.state_bytes == 25 * 4 and gen.voices == hand.voices
    assert gen.voices_per_lane == (2 if lanes == "two voices per lane" else 1) and hand.voices_per_lane == 2
    rng = np.random.default_rng(4)
    held = []
    def mirror(voices):
        gen.voices_upload(voices, np.stack([sub2a_to_graph(hand.voice_download(v)) for v in voices]))
    for block in range(40):
        if block % 3 == 0 and block < 24:                          # start a few notes
            started = []
            for _ in range(5):
                sy, pitch = int(rng.integers(0, S)), int(rng.integers(36, 97))
                slot = hand.note_on(sy, pitch, float(np.float32(rng.uniform(0.3, 1.0))))
                started.append(sy * P + slot); held.append((sy, pitch))
            mirror(sorted(set(started)))
        if block % 4 == 2 and held:                                # release some: off() runs on the resident state
            sy, pitch = held.pop(int(rng.integers(0, len(held))))
            before = hand.stages().copy()
            hand.note_off(sy, pitch)
            mirror([v for v in range(sy * P, (sy + 1) * P) if before[v] == 1])
        a, mix_a = hand.process_voices(N)
        b, mix_b = gen.process_voices(N)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"block {block}"
        assert np.allclose(mix_a, mix_b, rtol=0, atol=1e-5 * max(1.0, np.abs(mix_a).max()))   # summation order differs (two voices per lane vs one)
        assert np.array_equal(hand.stages(), gen.stages())
    assert (hand.stages() == 3).sum() > 0 and np.abs(a).max() > 0
    hand.close(); gen.close()


WAVETABLE_PROGRAM = """klgg 1
ctl 0
node 0 wavetable
node 1 param
op osc 0 -1 -1 0 0           # Wavetable::process
op param 1 -1 -1 1 0
op mul 2 0 1 -1 0             # wavetable * scale (a param member) ...
op tabread 3 2 -1 -1 2        # ... as a fractional index into table 2 (Table<float, SIZE>::operator[](float))
op add 4 0 3 -1 0
ret 4
end
"""


def test_graph_wavetable_and_table_read_match_numpy_model():
    """klg_table_upload + a wavetable node + a tabread op against the reference's arithmetic written out in numpy fp32:
    Phase += {increment, size}; buffer[float] = s[i] * (1 - frac) + s[j] * frac (klang.h:2070-2078, 3676-3679) and the clamped
    Table read s[i] + dx * (s[i + 1] - s[i]) (3365-3377).  Voices differ in table, increment, start position and scale."""
    import klang_amd
    f32 = np.float32
    V, N, B = 70, 64, 5                                   # two waves, the second partly filled
    bank = klang_amd.SynthBank(WAVETABLE_PROGRAM, synths=1, notes=V, max_block=N)
    assert bank.state_bytes == (1 + 5 + 1) * 4
    rng = np.random.default_rng(11)
    sine = np.sin(2 * np.pi * np.arange(2048) / 2048).astype(f32)
    shaper = np.tanh(np.linspace(-3, 3, 33)).astype(f32)
    odd = rng.uniform(-1, 1, 100).astype(f32)
    t_sine = bank.table_upload(sine)
    t_shaper = bank.table_upload(shaper, dedup=False)
    t_odd = bank.table_upload(odd)
    assert (t_sine, t_shaper, t_odd) == (1, 2, 3)
    assert bank.table_upload(sine.copy()) == 1 and bank.table_upload(sine.copy(), dedup=False) == 4   # same samples share an id unless asked not to
    tables = {1: sine, 3: odd}
    words = np.zeros((V, 7), np.uint32)
    state = []
    for v in range(V):
        tid = 1 if v % 3 else 3
        size = len(tables[tid])
        inc = f32(rng.uniform(0.3, 0.49 * size)) if v else f32(2.0 * size)   # voice 0: increment >= size never advances (klang.h:1528)
        pos = f32(rng.uniform(0, size - 1))
        scale = f32(rng.uniform(0, 48))
        words[v] = [1, inc.view(np.uint32), pos.view(np.uint32), 0, 0, tid, scale.view(np.uint32)]   # flags = Sustain
        state.append([inc, pos, tid, scale])
    bank.voices_upload(np.arange(V), words)
    for block in range(B):
        pv, _ = bank.process_voices(N)
        want = np.zeros((V, N), f32)
        for v, st in enumerate(state):
            inc, pos, tid, scale = st
            tab = tables[tid]; size = f32(len(tab))
            for i in range(N):
                if not inc >= size:
                    pos = f32(pos + inc)
                    if pos > size:
                        pos = f32(pos - size)
                fl = f32(np.floor(pos)); frac = f32(pos - fl)
                a = int(pos); a = a - len(tab) if a >= len(tab) else a
                b = 0 if a == len(tab) - 1 else a + 1
                y = f32(f32(tab[a] * f32(f32(1) - frac)) + f32(tab[b] * frac))
                idx = f32(y * scale)
                if idx < 0:
                    sh = shaper[0]
                elif idx >= len(shaper) - 1:
                    sh = shaper[-1]
                else:
                    x = f32(np.floor(idx)); k = int(x)
                    sh = f32(shaper[k] + f32(f32(idx - x) * f32(shaper[k + 1] - shaper[k])))
                want[v, i] = f32(y + sh)
            st[1] = pos
        assert np.array_equal(pv.view(np.uint32), want.view(np.uint32)), f"block {block}: max err {np.abs(pv - want).max()}"
    assert np.abs(pv).max() > 0.5
    bank.close()


CTL_PROGRAM = """klgg 1
ctl 2
dial 0 0 1 0.5
dial 1 0 2 1
node 0 saw
node 1 fsine
node 2 adsr
op osc 0 -1 -1 0 0
op osc 1 -1 -1 1 0
op ctl 2 -1 -1 -1 0
op ctl 3 -1 -1 -1 1
op mul 4 0 2 -1 0
op mul 5 1 3 -1 0
op add 6 4 5 -1 0
op env 7 -1 -1 2 0
op mul 8 6 7 -1 0
op stopif -1 -1 -1 2 0
ret 8
end
"""


def test_graph_two_voices_per_lane_reads_each_voices_own_controls(monkeypatch):
    """An odd number of notes per synth puts the two voices of a lane in DIFFERENT synth instances (different controls); duty != 0 on
    some voices keeps the general saw form in play; an ADSR that finishes mid-run exercises the packed segment ends and stop."""
    import klang_amd
    S, P, N = 5, 3, 96
    rng = np.random.default_rng(21)
    monkeypatch.setenv("KLG_GRAPH_SP", "0")                           # (a bank this small would take the sample-parallel form: this test is about the packed one)
    def make(env):
        if env:
            monkeypatch.setenv("KLG_GRAPH_X1", "1")
        else:
            monkeypatch.delenv("KLG_GRAPH_X1", raising=False)
        b = klang_amd.SynthBank(CTL_PROGRAM, synths=S, notes=P, max_block=N)
        for sy in range(S):
            b.set_control(sy, 0, 0.1 + 0.2 * sy); b.set_control(sy, 1, 1.9 - 0.3 * sy)
        return b
    one, two = make(True), make(False)
    assert one.voices_per_lane == 1 and two.voices_per_lane == 2
    f = np.float32
    words = np.zeros((S * P, 1 + 6 + 3 + 9), np.uint32)
    for v in range(S * P):
        inc = np.int32(rng.integers(1 << 20, 1 << 26)); delta = f((int(inc) >> 9 | 0x3f800000)); delta = np.uint32((int(inc) >> 9) | 0x3f800000).view(f) - f(1)
        duty = np.uint32(0 if v % 2 else rng.integers(1 << 28, 1 << 31))
        words[v, 0] = 1
        words[v, 1:7] = [np.uint32(inc), np.uint32(rng.integers(0, 1 << 32)), duty, delta.view(np.uint32), 0, 0]
        words[v, 7:10] = [np.uint32(rng.integers(1 << 20, 1 << 25)), np.uint32(rng.integers(0, 1 << 32)), 0]
        A, AD, Sus, R = f(0.0005 * (1 + v % 3)), f(0.0015 * (1 + v % 3)), f(0.6), f(0.001)
        # attack from 0: target 1 at time A (Envelope::setTargetTime: rate = |1 - 0| / ((A - 0) * fs)); ramp active, stage Sustain, point 0
        words[v, 10:19] = [f(0).view(np.uint32), f(1).view(np.uint32), f(1.0 / (float(A) * 48000.0)).view(np.uint32), f(0).view(np.uint32), (0 | (0 << 2) | (1 << 5)),
                           A.view(np.uint32), AD.view(np.uint32), Sus.view(np.uint32), R.view(np.uint32)]
    for b in (one, two):
        b.voices_upload(np.arange(S * P), words)
    for block in range(6):
        if block == 3:                                            # release half the voices the way ADSR::release does: stage Release, ramp to 0 over R
            for b in (one, two):
                for v in range(0, S * P, 2):
                    w = b.voice_download(v).copy()
                    out = w[10:11].view(f)[0]
                    w[11] = f(0).view(np.uint32); w[12] = f(abs(0 - out) / (0.001 * 48000.0)).view(np.uint32); w[14] = (1 | (int(w[14]) & 0x1C) | (1 << 5))
                    b.voice_upload(v, w)
        a, _ = one.process_voices(N)
        c, _ = two.process_voices(N)
        assert np.array_equal(a.view(np.uint32), c.view(np.uint32)), f"block {block}: max err {np.abs(a - c).max()}"
        assert np.array_equal(one.stages(), two.stages())
        for v in range(S * P):
            assert np.array_equal(one.voice_download(v), two.voice_download(v)), f"block {block} voice {v}"
    assert np.abs(a).max() > 0.01 and (one.stages() == 3).sum() >= S * P // 2
    per_synth = np.abs(a.reshape(S, P, N)).max(axis=(1, 2))
    assert len(set(np.round(per_synth, 3))) > 1                    # the synths' different controls are audible
    one.close(); two.close()


def test_table_and_delay_entry_points_reject_misuse():
    import klang_amd
    from klang_amd._lib import KlangError
    hand = klang_amd.SynthBank("sub2a", synths=1, notes=4, max_block=64)
    with pytest.raises(KlangError, match="only graph banks"):
        hand.table_upload(np.zeros(8, np.float32))
    assert hand._L.klg_voice_delay_clear(hand._h, 0, 0) < 0 and b"no note delays" in hand._L.klg_last_error()
    hand.close()
    g = klang_amd.SynthBank(WAVETABLE_PROGRAM, synths=1, notes=4, max_block=64)
    with pytest.raises(KlangError, match="bad arguments"):
        g.table_upload(np.zeros(1, np.float32))                     # a table needs two samples
    assert g._L.klg_voice_delay_clear(g._h, 0, 0) < 0               # this program has no delay
    assert g.voices_per_lane == 1                                   # wavetable nodes have no packed form
    g.close()


def test_graph_program_errors_are_reported():
    import klang_amd
    with pytest.raises(klang_amd.KlangError, match="operand a is not defined"):
        klang_amd.SynthBank("klgg 1\nctl 0\nnode 0 lpf\nop lpf 1 0 -1 0 0\nret 1\nend\n", synths=1, notes=1)


# ---- graph effects (`kind effect`): a hand-written program against a numpy model of the same arithmetic ----
ECHO_PROGRAM = """klgg 1
kind effect 1
ctl 2
dial 0 0.001 1 0.01
dial 1 0 1 0.5
node 0 delay 4800
op in 0 -1 -1 -1 0
op ctl 1 -1 -1 -1 0
op const 2 -1 -1 -1 473b8000      # 48000.f
op mul 3 1 2 -1 0                 # time = controls[0] * fs
op ctl 4 -1 -1 -1 1
op delayin -1 0 -1 0 0            # in >> delay
op delaytap 5 3 -1 0 0            # delay(time)
op mul 6 5 4 -1 0
op add 7 0 6 -1 0                 # in + delay(time) * gain >> out
ret 7
end
"""


def test_graph_effect_echo_matches_numpy_model():
    import klang_amd
    K, N, B, SIZE = 70, 96, 12, 4800
    bank = klang_amd.FxBank(ECHO_PROGRAM, K, max_block=N, channels=1)
    rng = np.random.default_rng(8)
    times = rng.uniform(0.002, 0.09, K).astype(np.float32); gains = rng.uniform(0, 1, K).astype(np.float32)
    for k in range(K):
        bank.set_control(k, 0, float(times[k])); bank.set_control(k, 1, float(gains[k]))
    x = rng.uniform(-0.5, 0.5, (B, K, 1, N)).astype(np.float32)
    ring = np.zeros((K, SIZE), np.float32); pos = 0
    for b in range(B):
        io = x[b].copy()
        bank.process(io)
        ref = np.zeros((K, N), np.float32)
        for i in range(N):                                           # Delay::input then tap(float) klang.h:3396-3427, fp32 throughout
            ring[:, pos] = x[b, :, 0, i]; pos = (pos + 1) % SIZE
            read = np.float32(pos - 1) - times * np.float32(48000.0)
            read = np.where(read < 0, read + np.float32(SIZE), read).astype(np.float32)
            ii = read.astype(np.int32); frac = (read - ii.astype(np.float32)).astype(np.float32); jj = (ii + 1) % SIZE
            a, c = ring[np.arange(K), ii], ring[np.arange(K), jj]
            ref[:, i] = x[b, :, 0, i] + (a + frac * (c - a)).astype(np.float32) * gains
        assert np.array_equal(io[:, 0].view(np.uint32), ref.view(np.uint32)), f"block {b}"
    bank.close()
