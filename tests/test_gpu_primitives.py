"""Every device primitive of the hot-path table on its own: klg_selftest() runs one
primitive in one GPU lane and the result is compared BIT FOR BIT with the reference's known-answer vectors
(tests/golden/prims.kat, produced by the genuine header; stimuli as in oracle/ref/ref_prims.cpp).
No tolerance anywhere: the swept biquad's cosf/sinf are a libm-exact restatement."""
import ctypes as C
import os

import numpy as np
import pytest

from scenario_io import fx_input, load_kat

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FS = 48000.0
FREQS = [27.5, 110.0, 440.0, 1000.0, 2093.0045, 7040.0, 15000.0]
(ST_BASIC_SINE, ST_BASIC_SAW, ST_BASIC_TRIANGLE, ST_BASIC_SQUARE, ST_BASIC_PULSE, ST_FAST_SINE, ST_OSM_SAW, ST_OSM_PULSE,
 ST_ONEPOLE_LPF, ST_ONEPOLE_HPF, ST_BIQUAD, ST_BIQUAD_LPF_SWEEP, ST_ADSR, ST_ENV3, ST_OPERATOR3, ST_DELAY, ST_STEREO_DELAY_TAP,
 ST_MATRIX, ST_CONTROL_SMOOTH, ST_NOISE_BASIC, ST_NOISE_FAST,
 ST_DCF, ST_IIR2, ST_IIR4, ST_IIR1, ST_BUTTER1, ST_MODAL, ST_FOLLOWER_AR, ST_FOLLOWER_PEAK, ST_FOLLOWER_RMS, ST_ENVN) = range(31)
F32P = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def kat():
    return load_kat(os.path.join(GOLDEN, "prims.kat"))


@pytest.fixture(scope="module")
def L():
    import klang_amd
    return klang_amd.lib()


def g(name):                     # %g formatting used for the KAT names
    return "%g" % name


def noise(i0, n):
    return fx_input(1, 0, 0, np.arange(i0, i0 + n, dtype=np.uint64), 0xFFFFFFFF)


def host(L, kind, args, n_out):
    a = np.asarray(args, np.float32)
    out = np.zeros(n_out, np.float32)
    rc = L.klg_selftest_host(kind, a.ctypes.data_as(F32P), len(a), C.c_float(FS), out.ctypes.data_as(F32P), n_out)
    assert rc == 0, L.klg_last_error()
    return out


def run(L, prim, params, n, inp=None, n_out=None):
    p = np.ascontiguousarray(params, np.float32)
    i = np.ascontiguousarray(inp if inp is not None else np.zeros(0), np.float32)
    n_out = n_out or n
    out = np.zeros(n_out, np.float32)
    rc = L.klg_selftest(prim, p.ctypes.data_as(F32P), len(p), i.ctypes.data_as(F32P), len(i), out.ctypes.data_as(F32P), n_out, n)
    assert rc == 0, L.klg_last_error()
    return out


def same(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    if a.shape != b.shape:
        return False
    # bit-for-bit, except that two NaNs compare equal whatever their payload (an all-pass of radius 2 diverges)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def test_pitch_to_frequency(L, kat):
    got = np.array([host(L, 8, [p], 1)[0] for p in range(128)], np.float32)
    assert same(got, kat["pitch_to_frequency"])


@pytest.mark.parametrize("f", FREQS)
def test_basic_oscillators(L, kat, f):
    inc, pos = host(L, 0, [f, 0.0], 2)
    for prim, tag in ((ST_BASIC_SINE, "sine"), (ST_BASIC_SAW, "saw"), (ST_BASIC_TRIANGLE, "triangle"), (ST_BASIC_SQUARE, "square")):
        assert same(run(L, prim, [inc, pos, 0.0, 0.5], 1024), kat[f"basic_{tag}_{g(f)}"]), tag
    assert same(run(L, ST_BASIC_PULSE, [inc, pos, 0.0, 0.25], 1024), kat[f"basic_pulse25_{g(f)}"])


def test_basic_sine_phase_and_relative_offset(L, kat):
    inc, pos = host(L, 0, [440.0, 1.5], 2)
    assert same(run(L, ST_BASIC_SINE, [inc, pos, 0.0, 0.5], 1024), kat["basic_sine_440_phase1.5"])
    inc, pos = host(L, 0, [440.0, 0.0], 2)
    two_pi = np.float32(2) * np.float32(np.pi)
    assert same(run(L, ST_BASIC_SINE, [inc, pos, np.float32(0.25) * two_pi, 0.5], 1024), kat["basic_sine_440_rel0.25"])


@pytest.mark.parametrize("f", FREQS)
def test_fast_sine_and_osm(L, kat, f):
    assert same(run(L, ST_FAST_SINE, host(L, 1, [f, 0.0], 2), 1024), kat[f"fast_sine_{g(f)}"])
    for tag, prim, ctor, args in (("saw", ST_OSM_SAW, 0.0, None), ("triangle", ST_OSM_SAW, 1.0, None), ("square", ST_OSM_PULSE, 1.0, None),
                                  ("pulse", ST_OSM_PULSE, 0.5, None), ("saw_duty0.05", ST_OSM_SAW, 0.0, (0.0, 0.05)), ("saw_phase1_duty0.615", ST_OSM_SAW, 0.0, (1.0, 0.615))):
        phase, duty, given = (args[0], args[1], 1.0) if args else (0.0, 0.0, 0.0)
        st = host(L, 2, [f, phase, duty, ctor, given], 5)
        assert same(run(L, prim, st, 1024), kat[f"fast_{tag}_{g(f)}"]), tag


def test_fast_sine_phase_modulation_with_negative_offsets(L, kat):
    """F3: float -> unsigned wrap of negative FM offsets (klang.h:4995-4996)."""
    assert same(run(L, ST_FAST_SINE, host(L, 1, [440.0, 0.0], 2), 1024, inp=np.float32(3.0) * noise(0, 1024)), kat["fast_sine_440_pm_noise3"])
    assert same(run(L, ST_FAST_SINE, host(L, 1, [440.0, 2.0], 2), 1024), kat["fast_sine_440_phase2"])


@pytest.mark.parametrize("f", FREQS)
def test_onepole_and_biquads(L, kat, f):
    x = noise(0, 1024)
    for kind, prim, tag in ((3, ST_ONEPOLE_LPF, "lpf"), (4, ST_ONEPOLE_HPF, "hpf")):
        c = host(L, kind, [f], 3)
        assert same(c, kat[f"onepole_{tag}_coef_{g(f)}"])
        assert same(run(L, prim, c, 1024, inp=x), kat[f"onepole_{tag}_{g(f)}"])
    for Q in (0.70710678, 0.3, 2.0, 10.0):
        for t, tag in enumerate(("lpf", "hpf", "bpf", "bpfskirt", "brf", "apf")):
            c = host(L, 5, [t, f, Q], 5)
            assert same(c, kat[f"biquad_{tag}_coef_{g(f)}_{g(np.float32(Q))}"]), (tag, Q)
            assert same(run(L, ST_BIQUAD, c, 256, inp=x[:256]), kat[f"biquad_{tag}_{g(f)}_{g(np.float32(Q))}"]), (tag, Q)


# ---- DCF, IIR<N>, IIR<1>, Butterworth::LPF<1>/<2>, Modal, Envelope::Follower ----
def test_f2_dcf_and_iir(L, kat):
    x = noise(0, 1024)
    assert same(run(L, ST_DCF, [0.995], 1024, inp=x), kat["dcf_default"])
    assert same(run(L, ST_DCF, [0.9], 1024, inp=x), kat["dcf_0.9"])
    assert same(run(L, ST_IIR2, [-1.2, 0.5], 1024, inp=x), kat["iir2"])
    assert same(run(L, ST_IIR4, [-0.5, 0.25, -0.125, 0.0625], 1024, inp=x), kat["iir4"])
    a = np.float32(0.25)
    assert same(run(L, ST_IIR1, [a, np.float32(1.0) - a], 1024, inp=x), kat["iir1_0.25"])


@pytest.mark.parametrize("f", FREQS)
def test_f2_butterworth(L, kat, f):
    x = noise(0, 1024)
    c = host(L, 9, [f], 2)
    assert same(c, kat[f"butter1_coef_{g(f)}"])
    assert same(run(L, ST_BUTTER1, c, 1024, inp=x), kat[f"butter1_{g(f)}"])
    c = host(L, 10, [f], 5)
    assert same(c, kat[f"butter2_coef_{g(f)}"])
    assert same(run(L, ST_BIQUAD, c, 1024, inp=x), kat[f"butter2_{g(f)}"])          # LPF<2> is a Biquad::Filter with its own init()


def test_f2_modal_and_follower(L, kat):
    i = np.arange(1024)
    x = np.where(i % 97 == 0, np.float32(1.0), np.float32(0.25) * noise(0, 1024)).astype(np.float32)
    for k, args in enumerate(([440.0, 0.5, 0.0], [1000.0, 0.05, 0.0], [110.0, 2.0, 0.5])):
        c = host(L, 11, args, 3)
        assert same(c, kat[f"modal_coef_{k}"]), k
        assert same(run(L, ST_MODAL, c, 1024, inp=x), kat[f"modal_{k}"]), k
    c = host(L, 12, [0.01, 0.1], 2)
    assert same(c, kat["follower_ar_coef"])
    gate = np.where((i // 200) % 2 == 1, np.float32(0.1), np.float32(1.0)).astype(np.float32)
    assert same(run(L, ST_FOLLOWER_AR, c, 1024, inp=np.abs(noise(0, 1024)) * gate), kat["follower_ar"])
    assert same(run(L, ST_FOLLOWER_PEAK, c, 1024, inp=noise(0, 1024) * gate), kat["follower_peak"])
    assert same(run(L, ST_FOLLOWER_RMS, c, 1024, inp=noise(0, 1024) * gate), kat["follower_rms"])


def test_biquad_swept_cutoff_on_device(L, kat):
    """The F6 path: coefficients recomputed every sample on the GPU with the libm-exact cosf/sinf restatement
    (glibc 2.35 FMA variant, klg_device.hpp: glibc_sincosf) — bit-exact like everything else."""
    fc = (np.float32(500.0) + np.float32(7.0) * np.arange(1024, dtype=np.float32)).astype(np.float32)
    w = np.float32(2.0) * np.float32(np.pi) * (np.float32(1.0) / np.float32(FS))
    got = run(L, ST_BIQUAD_LPF_SWEEP, [10.0, w], 1024, inp=np.concatenate([noise(0, 1024), fc]))
    assert same(got, kat["biquad_lpf_sweep_q10"])


def test_adsr_and_envelopes(L, kat):
    def adsr(a, d, s, r, n, release_at=-1, rel_time=0.0, rel_level=0.0):
        st = host(L, 6, [a, d, s, r], 9)
        return run(L, ST_ADSR, list(st) + [release_at, rel_time, rel_level, FS], n, n_out=2 * n)
    assert same(adsr(1e-4, 1e-4, 0.5, 1e-4, 64)[:64], kat["adsr_1e-4"])
    o = adsr(0.01, 0.1, 0.7, 0.25, 24000, release_at=9000)
    assert same(o[:24000:8], kat["adsr_std_release9000_dec8"]) and same(o[24000::8], kat["adsr_std_release9000_stage_dec8"])
    assert same(o[:1024], kat["adsr_std_release9000_head"]) and same(o[8990:9100], kat["adsr_std_release9000_rel"])
    assert same(adsr(0.0, 0.0, 1.0, 0.25, 2000, release_at=1000)[:2000], kat["adsr_0_0_1_release1000"])
    assert same(adsr(0.001, 0.25, 1.0, 0.5, 2000, release_at=20)[:2000], kat["adsr_release_during_attack"])
    assert same(adsr(0.01, 0.1, 0.7, 0.25, 3000, release_at=100, rel_time=0.01, rel_level=0.2)[:3000], kat["adsr_release_time_level"])

    def env(points, n):
        flat = [v for p in points for v in p]
        st = host(L, 7, [len(points)] + flat, 5)
        xs = [p[0] for p in points] + [0.0] * (3 - len(points)); ys = [p[1] for p in points] + [0.0] * (3 - len(points))
        return run(L, ST_ENV3, list(st) + [len(points)] + xs + ys + [FS], n, n_out=2 * n)
    o = env([(0, 880), (0.01, 4400), (0.03, 2200)], 2048)
    assert same(o[:2048], kat["envelope_3pt"]) and same(o[2048:], kat["envelope_3pt_stage"])
    o = env([(0, 1)], 16)
    assert same(o[:16], kat["envelope_default"]) and same(o[16:], kat["envelope_default_stage"])
    assert same(env([(0, 1.5), (3, 0.5)], 1024)[:1024], kat["envelope_fm_op2"])


def test_envelopes_of_any_length_loops_and_rate_mode(L, kat):
    """SURVEY row a16 (klang.h:3812, 3893-3909, 3923-3950, 4031, 4064-4092): the run-time envelope of recorded graph patches — env_process_rt over
    PtsN: four points in registers, the others read from the record when a segment ends — against the genuine header's vectors: more than four
    points, loops over later points, resetLoop(), Rate mode (a jump point, a loop), release() in either mode.  Two record capacities each."""
    def envn(points, n, rate=False, loop=(-1, -1), release=(-1, 0.0, 0.0), cap=None):
        cap = cap or max(5, len(points))
        flat = [v for p in points for v in p]
        st = host(L, 13, [1.0 if rate else 0.0, loop[0], loop[1], len(points)] + flat, 5)
        xs = [p[0] for p in points] + [0.0] * (cap - len(points)); ys = [p[1] for p in points] + [0.0] * (cap - len(points))
        words = xs[:4] + ys[:4] + xs[4:] + ys[4:]

        def go(state, n_, ls, le, rel):
            return run(L, ST_ENVN, list(state) + [len(points), ls, le, rel[0], rel[1], rel[2], FS, cap], n_, inp=words, n_out=2 * n_)
        return go(st, n, loop[0], loop[1], release)
    for cap in (None, 16):
        o = envn([(0, 0.5), (0.002, 1), (0.004, 0), (0.006, 0.7), (0.008, 0.2), (0.01, 0.9), (0.012, 0.1), (0.014, 0.6), (0.016, 0.3), (0.03, 0)], 2048, cap=cap)
        assert same(o[:2048], kat["envelope_10pt"]) and same(o[2048:], kat["envelope_10pt_stage"])
        o = envn([(0, 0), (0.004, 1), (0.009, 0.3), (0.013, 0.8), (0.02, 0.1), (0.024, 0.6)], 2048, loop=(5, 5), release=(1500, 0.004, 0.05), cap=cap)
        assert same(o[:2048], kat["envelope_6pt_hold_5_release"]) and same(o[2048:], kat["envelope_6pt_hold_5_release_stage"])
        o = envn([(0, 0), (0.002, 1), (0, 0.25), (0.001, 0.75), (0.0005, 0.5), (0.004, 0)], 4096, rate=True, cap=cap)
        assert same(o[:4096], kat["envelope_rate_6pt_jump"]) and same(o[4096:], kat["envelope_rate_6pt_jump_stage"])
        o = envn([(0, 0), (0.002, 1), (0.001, 0.25), (0.003, 0.75), (0.0005, 0.5)], 4096, rate=True, loop=(1, 3), release=(3000, 0.0007, 0.1), cap=cap)
        assert same(o[:4096], kat["envelope_rate_loop_1_3_release"]) and same(o[4096:], kat["envelope_rate_loop_1_3_release_stage"])
        # the seven-point loop 2 .. 5 up to the sample where the reference lifts the loop (resetLoop at 6000 is host code: the note-level fixture own_env_points covers it)
        o = envn([(0, 0), (0.004, 1), (0.009, 0.3), (0.013, 0.8), (0.02, 0.1), (0.024, 0.6), (0.05, 0)], 6000, loop=(2, 5), cap=cap)
        assert same(o[:6000], kat["envelope_7pt_loop_2_5"][:6000]) and same(o[6000:], kat["envelope_7pt_loop_2_5_stage"][:6000])
    # the older vectors through the run-time form too: three points, the four-point loop 1 .. 3, the Rate-mode KAT of round 1 (which nothing ran on the device until now)
    o = envn([(0, 880), (0.01, 4400), (0.03, 2200)], 2048)
    assert same(o[:2048], kat["envelope_3pt"]) and same(o[2048:], kat["envelope_3pt_stage"])
    assert same(envn([(0, 0), (0.005, 1), (0.01, 0.25), (0.02, 0.5)], 4096, loop=(1, 3))[:4096], kat["envelope_loop_1_3"])
    assert same(envn([(0, 0), (0.001, 1), (0.0005, 0.2)], 4096, rate=True)[:4096], kat["envelope_rate_mode"])


def test_operator_chain(L, kat):
    p = []
    for f, pts, amp in ((220.0, [(0, 0), (3, 1)], 3.7), (220.0, [(0, 1.5), (3, 0.5)], 1.37), (440.0, [(0, 1)], 1.0)):
        osc = host(L, 1, [f, 0.0], 2)
        st = host(L, 7, [len(pts)] + [v for q in pts for v in q], 5)
        xs = [q[0] for q in pts] + [0.0] * (2 - len(pts)); ys = [q[1] for q in pts] + [0.0] * (2 - len(pts))
        p += list(osc) + list(st) + [len(pts)] + xs + ys + [amp]
    assert same(run(L, ST_OPERATOR3, p + [FS], 1024), kat["operator_chain3"])


def test_delay_taps(L, kat):
    ramp = np.arange(1, 41, dtype=np.float32)
    assert same(run(L, ST_DELAY, [16, 0, 3.5], 40, inp=ramp), kat["delay16_set3.5"])
    sq = np.array([(i * i) % 17 for i in range(1, 41)], np.float32)
    assert same(run(L, ST_DELAY, [16, 1, 5], 40, inp=sq), kat["delay16_tap_int5"])
    assert same(run(L, ST_DELAY, [16, 2, 2.25], 40, inp=sq), kat["delay16_tap_2.25"])
    assert same(run(L, ST_DELAY, [16, 3, 3.6], 40, inp=sq), kat["delay16_lagrange_3.6"])
    times = (np.float32(100.0) + np.float32(50.0) * noise(0, 3000)).astype(np.float32)
    assert same(run(L, ST_DELAY, [1000, 4, 0], 3000, inp=np.concatenate([noise(7777, 3000), times])), kat["delay1000_modulated_set"])
    assert same(run(L, ST_DELAY, [100, 4, 0], 400, inp=np.concatenate([noise(0, 400), np.full(400, 33.25, np.float32)])), kat["delay0_100_set33.25"])
    got = run(L, ST_STEREO_DELAY_TAP, [64, 10.75], 200, inp=np.concatenate([noise(0, 200), noise(5000, 200)]), n_out=400)
    assert same(got, kat["stereo_delay64_tap10.75"])


def test_matrix_control_noise(L, kat):
    assert same(run(L, ST_MATRIX, [0], 16, inp=noise(0, 64), n_out=64), kat["matrix_fdn"])
    assert same(run(L, ST_CONTROL_SMOOTH, [0.5], 512), kat["control_smooth_0.5"][:512])
    libc = C.CDLL(None)
    libc.srand(1)
    r = np.array([libc.rand() for _ in range(256)], np.uint32).view(np.float32)        # the injected rand() stream (F5)
    assert same(run(L, ST_NOISE_BASIC, [0], 256, inp=r), kat["basic_noise_srand1"])
    assert same(run(L, ST_NOISE_FAST, [0], 256, inp=r), kat["fast_noise_srand1"])
