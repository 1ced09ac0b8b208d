"""The headless host (klang_amd/host/klang_render.cpp): Standard MIDI File -> note events -> GPU blocks -> WAV.
CPU: the MIDI and WAV parsers against files made here with plain struct packing / the stdlib `wave` module.
GPU: a two-track MIDI file with a tempo change rendered through tests/patches/sub2a.k (recorded graph patch) equals the
oracle's render of the same events quantised to blocks the way the reference's processBlock does."""
import os
import struct
import subprocess
import wave

import numpy as np
import pytest

from scenario_io import Scenario

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "_bin")


def vlq(n):
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(out))


def track(events):
    """events: (absolute tick, raw bytes of the event after the delta time)"""
    data, t = b"", 0
    for tick, raw in sorted(events, key=lambda e: e[0]):
        data += vlq(tick - t) + raw
        t = tick
    data += vlq(0) + b"\xff\x2f\x00"
    return b"MTrk" + struct.pack(">I", len(data)) + data


def smf(tracks, division=480, fmt=1):
    return b"MThd" + struct.pack(">IHHH", 6, fmt, len(tracks), division) + b"".join(tracks)


def host(name):
    path = os.path.join(BIN, name)
    if not os.path.exists(path):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), os.path.join("_bin", name)], check=True)
    return path


def song():
    """(midi bytes, [(seconds, status, d1, d2)]) — tempo 120 -> 240 bpm at tick 960, running status, a sysex and a text meta in between"""
    tempo = track([(0, b"\xff\x51\x03\x07\xa1\x20"), (960, b"\xff\x51\x03\x03\xd0\x90")])
    notes = [(0, 60, 100), (240, 64, 80), (480, 67, 90), (1200, 72, 127), (1500, 48, 60)]
    ev = []
    for tick, pitch, vel in notes:
        ev.append((tick, bytes([0x90, pitch, vel])))
        ev.append((tick + 400, bytes([0x80, pitch, 0])))
    ev.append((100, b"\xf0\x03\x01\x02\xf7"))                      # sysex: skipped
    ev.append((300, b"\xff\x01\x02hi"))                            # text meta: skipped
    ev.append((700, bytes([0xB0, 7, 64])))                         # controller 7
    lead = track(ev)
    # second track: running status (two note-ons share one status byte), note-off as note-on with velocity 0
    bass = b"MTrk"
    body = vlq(120) + bytes([0x91, 36, 70]) + vlq(60) + bytes([40, 75]) + vlq(500) + bytes([36, 0]) + vlq(10) + bytes([40, 0]) + vlq(0) + b"\xff\x2f\x00"
    bass += struct.pack(">I", len(body)) + body

    def seconds(tick):
        return tick * 0.5 / 480 if tick <= 960 else 1.0 + (tick - 960) * 0.25 / 480
    want = []
    for tick, raw in ev:
        if raw[0] in (0x90, 0x80, 0xB0):
            want.append((tick, 1, seconds(tick), raw[0], raw[1], raw[2]))
    for tick, st, d1, d2 in [(120, 0x91, 36, 70), (180, 0x91, 40, 75), (680, 0x91, 36, 0), (690, 0x91, 40, 0)]:
        want.append((tick, 2, seconds(tick), st, d1, d2))
    want.sort(key=lambda e: (e[0], e[1]))
    return smf([tempo, lead, bass]), [(s, st, a, b) for _, _, s, st, a, b in want]


def test_midi_file_is_parsed_into_time_stamped_messages(tmp_path):
    data, want = song()
    mid = tmp_path / "song.mid"
    mid.write_bytes(data)
    out = subprocess.run([host("klang_render"), str(mid), "--events"], capture_output=True, text=True, check=True).stdout.splitlines()
    assert out[0] == f"format 1 tracks 3 division 480 events {len(want)}"
    got = [(float(l.split()[0]), int(l.split()[3], 16), int(l.split()[4]), int(l.split()[5])) for l in out[1:]]
    assert [g[1:] for g in got] == [w[1:] for w in want]
    assert np.allclose([g[0] for g in got], [w[0] for w in want], rtol=0, atol=1e-9)


def test_smpte_division_and_errors(tmp_path):
    mid = tmp_path / "smpte.mid"
    mid.write_bytes(smf([track([(0, bytes([0x90, 60, 1])), (3000, bytes([0x80, 60, 0]))])], division=0xE728, fmt=0))   # 25 fps x 40 ticks/frame
    out = subprocess.run([host("klang_render"), str(mid), "--events"], capture_output=True, text=True, check=True).stdout.splitlines()
    assert float(out[2].split()[0]) == pytest.approx(3.0)
    bad = tmp_path / "bad.mid"
    bad.write_bytes(b"RIFFxxxxWAVE")
    r = subprocess.run([host("klang_render"), str(bad), "--events"], capture_output=True, text=True)
    assert r.returncode == 1 and "not a Standard MIDI File" in r.stderr
    trunc = tmp_path / "trunc.mid"
    trunc.write_bytes(song()[0][:-20])
    r = subprocess.run([host("klang_render"), str(trunc), "--events"], capture_output=True, text=True)
    assert r.returncode == 1 and "truncated" in r.stderr


def test_wav_reader_decodes_pcm16_stereo(tmp_path):
    rng = np.random.default_rng(3)
    pcm = rng.integers(-32768, 32767, size=(1000, 2), dtype=np.int16)
    path = tmp_path / "a.wav"
    with wave.open(str(path), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(22050); w.writeframes(pcm.tobytes())
    out = subprocess.run([host("klang_render"), "--wav-info", str(path)], capture_output=True, text=True, check=True).stdout.splitlines()
    assert out[0] == "rate 22050 channels 2 frames 1000"
    for c in range(2):
        f = pcm[:, c].astype(np.float32) / np.float32(32768)
        toks = out[1 + c].split()
        assert float(toks[3]) == pytest.approx(float(f.astype(np.float64).sum()), rel=1e-6, abs=1e-6)
        assert float(toks[5]) == pytest.approx(float(np.abs(f).max()), rel=1e-6)
        assert float(toks[7]) == pytest.approx(float(f[0]), rel=1e-6, abs=1e-9)


def test_wav_reader_decodes_float32_and_pcm24(tmp_path):
    x = np.linspace(-0.9, 0.9, 50).astype(np.float32)
    def riff(fmt, bits, payload, ch=1, rate=48000):
        hdr = b"WAVEfmt " + struct.pack("<IHHIIHH", 16, fmt, ch, rate, rate * ch * bits // 8, ch * bits // 8, bits) + b"LIST" + struct.pack("<I", 4) + b"abcd" + b"data" + struct.pack("<I", len(payload)) + payload
        return b"RIFF" + struct.pack("<I", len(hdr)) + hdr
    f32 = tmp_path / "f32.wav"
    f32.write_bytes(riff(3, 32, x.tobytes()))                                  # IEEE float, with a LIST chunk before the data
    out = subprocess.run([host("klang_render"), "--wav-info", str(f32)], capture_output=True, text=True, check=True).stdout.splitlines()
    assert out[0] == "rate 48000 channels 1 frames 50" and float(out[1].split()[7]) == pytest.approx(-0.9, rel=1e-6)
    q = np.round(x.astype(np.float64) * 8388607).astype(np.int32)
    p24 = tmp_path / "p24.wav"
    p24.write_bytes(riff(1, 24, b"".join(int(v).to_bytes(3, "little", signed=True) for v in q)))
    out = subprocess.run([host("klang_render"), "--wav-info", str(p24)], capture_output=True, text=True, check=True).stdout.splitlines()
    assert out[0] == "rate 48000 channels 1 frames 50"
    assert float(out[1].split()[7]) == pytest.approx(q[0] / 8388608.0, rel=1e-6) and float(out[1].split()[5]) == pytest.approx(np.abs(q).max() / 8388608.0, rel=1e-6)


@pytest.mark.parametrize("name", ["u8", "s16", "s32", "f32"])
def test_wav_reader_matches_the_reference_decoder(name):
    """include/klang/host/wav.hpp against File::WAV of the genuine header (klang.h:5991-6099) on every encoding that decoder knows:
    the committed files of tests/golden/wav/ were decoded by oracle/_ref/ref_wav (oracle/gen_golden_wav.py) — bit for bit."""
    golden = os.path.join(ROOT, "tests", "golden")
    want = np.load(os.path.join(golden, "wav_expected.npz"))[name]
    lines = subprocess.run([host("klang_render"), "--wav-dump", os.path.join(golden, "wav", name + ".wav")], capture_output=True, text=True, check=True).stdout.split()
    got = np.array([int(x, 16) for x in lines[1:]], dtype=np.uint32)
    assert int(lines[0]) == len(want) == len(got)
    assert np.array_equal(got, want)


def read_wav_f32(path):
    d = open(path, "rb").read()
    assert d[:4] == b"RIFF" and d[8:16] == b"WAVEfmt "
    fmt, ch, rate = struct.unpack("<HHI", d[20:28])
    assert fmt == 3 and d[36:40] == b"data"
    n = struct.unpack("<I", d[40:44])[0]
    return rate, np.frombuffer(d, np.float32, n // 4, 44).reshape(-1, ch).T


@pytest.mark.gpu
def test_midi_to_wav_matches_the_oracle_render_of_the_same_events(tmp_path, oracle_build):
    from klg_driver import run_scenario_oracle
    data, events = song()
    mid, wav = tmp_path / "song.mid", tmp_path / "song.wav"
    mid.write_bytes(data)
    fs, N, tail = 48000.0, 256, 0.5
    r = subprocess.run([host("klang_render_sub2a"), str(mid), str(wav), "--fs", "48000", "--block", str(N), "--tail", str(tail)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rate, audio = read_wav_f32(str(wav))
    blocks = audio.shape[1] // N
    assert rate == 48000 and audio.shape[0] == 2 and blocks == int(np.ceil(np.ceil((events[-1][0] + tail) * fs) / N))
    # the same events, quantised to blocks like processBlock: everything stamped before a block's end is delivered before it
    s = Scenario(patch="sub2a", fs=fs, block=N, blocks=blocks, synths=1, notes=16, dump=[])
    for sec, st, d1, d2 in events:
        b = int(np.floor(sec * fs)) // N
        if st & 0xF0 == 0x90 and d2 > 0:
            s.on(b, 0, d1, np.float32(d2) / np.float32(127))
        elif st & 0xF0 == 0x80 or (st & 0xF0 == 0x90 and d2 == 0):
            s.off(b, 0, d1, np.float32(d2) / np.float32(127))
    ref = run_scenario_oracle(s, oracle_build)["mix"]                # [blocks][2][N]
    got = audio.reshape(2, blocks, N).transpose(1, 0, 2)
    peak = np.abs(ref).max()
    assert peak > 0.1
    assert np.abs(got - ref).max() <= 1e-5 * peak                     # voices are summed in another order than on the CPU
    assert np.abs(got[-1]).max() < 1e-3                               # the tail has rung out


@pytest.mark.gpu
def test_controllers_and_initial_controls_reach_the_patch(tmp_path):
    """tests/patches/branches.k compares controls[0] with 0.5 per sample: `--control 0=0.2` sets it before the first block, MIDI
    controller 7 (mapped with `--cc 7=0`) moves it to 64/127 mid-song.  Checked against the GENUINE reference header running the
    same patch on the same events (oracle/_ref/ref_own_branches; built where the reference exists, travels with the snapshot)."""
    from scenario_io import load_ref_output
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "ref_own_branches")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref/ref_own_branches is built only where the reference header exists")
    data, events = song()
    mid, wav = tmp_path / "song.mid", tmp_path / "song.wav"
    mid.write_bytes(data)
    fs, N, tail = 48000.0, 256, 0.5
    r = subprocess.run([host("klang_render_branches"), str(mid), str(wav), "--block", str(N), "--tail", str(tail), "--control", "0=0.2", "--cc", "7=0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    _, audio = read_wav_f32(str(wav))
    blocks = audio.shape[1] // N
    s = Scenario(patch="own_branches", fs=fs, block=N, blocks=blocks, synths=1, notes=16, dump=[])
    s.ctl.append((0, 0.2))
    for sec, st, d1, d2 in events:
        b = int(np.floor(sec * fs)) // N
        if st & 0xF0 == 0x90 and d2 > 0:
            s.on(b, 0, d1, np.float32(d2) / np.float32(127))
        elif st & 0xF0 == 0x80 or (st & 0xF0 == 0x90 and d2 == 0):
            s.off(b, 0, d1, np.float32(d2) / np.float32(127))
        elif st & 0xF0 == 0xB0 and d1 == 7:
            s.control(b, 0, 0, np.float32(d2) / np.float32(127) * np.float32(1.0) + np.float32(0.0))   # Control::setNormalised: norm * range + min
    scn, out = tmp_path / "s.scn", tmp_path / "ref.bin"
    s.save(str(scn))
    subprocess.run([ref_bin, str(scn), str(out)], check=True)
    ref = load_ref_output(str(out))["mix"][:, 0, :]                 # mono synth
    got = audio[0].reshape(blocks, N)
    peak = np.abs(ref).max()
    assert peak > 0.1 and np.abs(got - ref).max() <= 1e-5 * peak


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--spin", "1"], ["--paced", "1"]])
def test_deadline_host_reports_every_block_and_whose_the_worst_one_was(extra):
    """klang_amd/host/klang_deadline.cpp (bench.py's `c_host` legs): a small bank, every block between two HIP events — the record says how many blocks were over 90 % of the
    deadline, how many of those were the device's own, and (paced: against the audio clock) how many blocks of buffering had no gap."""
    import json
    exe = os.path.join(ROOT, "klang_amd", "host", "klang_deadline")
    assert os.path.exists(exe), "klang_amd/csrc/build.sh builds it"
    p = subprocess.run([exe, "--voices", "65536", "--blocks", "200"] + extra, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-500:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["voices"] == 65536 and d["blocks"] == 200 and d["finite"] is True
    assert len(d["ten_largest"]) == 10 and len(d["ten_largest_device_ms"]) == 10 and len(d["ten_largest_queueing_ms"]) == 10
    assert 0 < d["device_p50_ms"] <= d["p50_ms"] < d["deadline_ms"]         # 65,536 voices are far inside the deadline; the device's share is inside the wall clock
    assert 0 <= d["of_them_the_devices"] <= d["blocks_over_90_percent"] <= 200
    assert ("polling" in d["host"]) == ("--spin" in extra)
    if "--paced" in extra:
        pc = d["paced"]
        assert abs(pc["period_ms"] - 256e3 / 48000) < 1e-3 and len(pc["blocks_later_than_1_2_3_periods"]) == 3 and 1 <= pc["buffering_with_no_gap_in_this_run_blocks"] <= 4
    else:
        assert "paced" not in d
