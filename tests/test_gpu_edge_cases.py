"""Edge cases of the block protocol on the GPU, checked against the TEST-ONLY oracle (ctypes, bit-for-bit):
ragged block lengths (1 .. max_block, not multiples of the 16/32-sample mix chunks), voice counts that are not
multiples of a wave / workgroup / the 2-voices-per-lane pairing, note-on and note-off inside the same block,
note-off for a pitch that is not sounding, control changes between blocks, an all-silent bank."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ko(oracle_build):
    L = C.CDLL(os.path.join(oracle_build, "libklang_oracle.so"))
    L.ko_bank_create.restype = C.c_void_p
    L.ko_bank_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float]
    L.ko_bank_note_on.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_long]
    L.ko_bank_note_off.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float]
    L.ko_bank_control.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float]
    L.ko_bank_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.ko_patch_from_name.argtypes = [C.c_char_p]
    return L


class Pair:
    """The same event stream into the GPU bank and the oracle bank."""
    def __init__(self, ko, patch, synths, notes, max_block=1024):
        import klang_amd
        self.ko, self.V = ko, synths * notes
        self.g = klang_amd.SynthBank(patch, synths=synths, notes=notes, max_block=max_block)
        self.o = ko.ko_bank_create(ko.ko_patch_from_name(patch.encode()), synths, notes, C.c_float(48000.0))

    def on(self, s, p, v=0.8, seed=-1):
        if seed >= 0:
            self.g.random(seed)
        a = self.g.note_on(s, p, v)
        b = self.ko.ko_bank_note_on(self.o, s, p, C.c_float(v), seed)
        assert a == b, "voice allocation differs"

    def off(self, s, p):
        self.g.note_off(s, p)
        self.ko.ko_bank_note_off(self.o, s, p, C.c_float(0.0))

    def ctl(self, s, i, v):
        self.g.set_control(s, i, v)
        self.ko.ko_bank_control(self.o, s, i, C.c_float(v))

    def block(self, n):
        pv, mix = self.g.process_voices(n)
        ref = np.zeros((self.V, n), np.float32); st = np.zeros(self.V, np.uint8)
        self.ko.ko_bank_process(self.o, ref.ctypes.data_as(C.c_void_p), None, st.ctypes.data_as(C.c_void_p), n)
        same = (pv.view(np.uint32) == ref.view(np.uint32)) | ((pv == 0) & (ref == 0))
        assert same.all(), f"n={n}: {int((~same).sum())} samples differ, max |d| {np.abs(pv - ref).max():.3e}"
        assert np.array_equal(self.g.stages(), st)
        tot = ref.astype(np.float64).sum(axis=0)
        assert np.max(np.abs(mix[0] - tot)) <= 1e-5 * max(1e-9, np.abs(ref).max()) * np.sqrt(self.V) * 4
        return pv

    def close(self):
        self.g.close()


RAGGED = [1, 7, 100, 257, 1024, 33, 16, 15, 31, 32, 2]


@pytest.mark.parametrize("patch,synths,notes,seeded", [("sub2a", 3, 100, False), ("sub2a", 1, 1, False), ("supersaw", 5, 32, True), ("fm4", 3, 32, False), ("sub2b", 2, 32, False), ("sine", 1, 5, False)])
def test_ragged_blocks_and_odd_voice_counts(ko, patch, synths, notes, seeded):
    p = Pair(ko, patch, synths, notes)
    rng = np.random.default_rng(11)
    for s in range(synths):
        for k in range(max(1, notes - 3)):                 # leave a few slots Off
            p.on(s, int(rng.integers(36, 97)), 0.7, int(rng.integers(1, 1 << 30)) if seeded else -1)
    for i, n in enumerate(RAGGED):
        if i == 4:
            for s in range(synths):
                p.off(s, 60); p.off(s, int(rng.integers(36, 97)))
        p.block(n)
    p.close()


def test_on_and_off_inside_one_block_and_unknown_pitch(ko):
    p = Pair(ko, "sub2a", 1, 8, max_block=256)
    p.on(0, 60); p.off(0, 60)            # released before it ever rendered a sample
    p.on(0, 64)
    p.off(0, 99)                         # nothing sounding at 99: no-op
    pv = p.block(256)
    # voice 0 was released at envelope value 0: release() targets 0 from 0, the ramp is idle and the note ends silently
    assert not np.any(pv[0]) and np.any(pv[1] != 0) and not np.any(pv[2:])
    p.off(0, 64); p.on(0, 64)            # release + retrigger in the same gap: a new slot, the old one keeps releasing
    for _ in range(3):
        p.block(256)
    p.close()


def test_control_changes_between_blocks(ko):
    p = Pair(ko, "fm4", 2, 32, max_block=128)
    for s in range(2):
        for k in range(10):
            p.on(s, 40 + 3 * k)
    p.block(128)
    p.ctl(0, 1, 4.5); p.ctl(1, 2, 0.0); p.ctl(0, 3, 9.9)      # operator amps are read every block on the GPU
    p.block(128)
    p.ctl(1, 0, 2.0); p.ctl(1, 4, 0.9)                        # Mod Freq / Attack only affect notes started afterwards
    p.on(1, 77)
    p.block(128)
    p.close()


def test_silent_bank_and_max_block():
    import klang_amd
    b = klang_amd.SynthBank("sub2a", synths=7, notes=128, max_block=1024)
    out = np.full((2, 1024), 0.25, np.float32)
    b.process(out)                                            # nothing sounding: the caller's buffer is left untouched (+= 0)
    assert np.all(out == 0.25) and np.all(b.stages() == 3)
    b.close()


def test_script_outlives_its_bank_without_touching_freed_state():
    """a klg_script keeps a pointer to its bank.  Destroying the bank first invalidates the script (its device arrays go with the
    bank); every later klg_script_* call fails with an error instead of dereferencing freed memory, and klg_script_destroy still works."""
    import torch
    import klang_amd
    bank = klang_amd.SynthBank("sub2a", synths=2, notes=8, max_block=64)
    script = klang_amd.EventScript(bank, 4)
    first = script.add_records(bank.note_records([0, 1], [60, 64], [0.8, 0.7]))
    script.note_on([0, 1], [0, 9], [first, first + 1])
    script.commit()
    mix = torch.zeros((2, 64), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()                              # (no stream given = the bank's own, non-blocking stream: torch's fill on the default stream is not ordered with it)
    script.play_device(0, mix.data_ptr(), 64)
    bank.sync()
    assert float(mix.abs().sum()) > 0
    bank.close()                                          # the bank goes first
    with pytest.raises(klang_amd.KlangError, match="destroyed"):
        script.play_device(1, mix.data_ptr(), 64)
    script.close()                                        # and the handle can still be released


@pytest.mark.parametrize("patch,synths,notes", [("sub2a", 4, 128), ("supersaw", 3, 32), ("fm4", 5, 32), ("sub2b", 2, 32)])
def test_single_launch_blocks_equal_the_separate_launches(patch, synths, notes):
    """Banks of a few workgroups render a block in ONE launch: the workgroups apply the block's note events of their own voices, the last one to
    finish adds the partial rows to the mix (RenderArgs::ev / ticket).  KLG_FUSE=0 selects the separate klg_apply_events / klg_render / klg_reduce
    launches: same per-voice samples and note stages bit for bit, the mix within the summation-order bound (rows are added in another order),
    and the single-launch mix is bit-reproducible from run to run."""
    import os
    from klg_driver import run_scenario_gpu
    from scenario_io import Scenario
    s = Scenario(patch=patch, block=192, blocks=10, synths=synths, notes=notes, dump=list(range(10)))
    rng = np.random.default_rng(5)
    for b in range(8):
        for _ in range(6):
            sy, p = int(rng.integers(0, synths)), int(rng.integers(40, 90))
            s.on(b, sy, p, float(rng.uniform(0.3, 1.0)), seed=int(rng.integers(1, 1 << 30)))
            s.off(b + 2, sy, p)
    s.sort()
    try:
        os.environ["KLG_FUSE"] = "0"
        ref = run_scenario_gpu(s)
        os.environ["KLG_FUSE"] = "1"
        got = run_scenario_gpu(s)
        again = run_scenario_gpu(s)
    finally:
        os.environ.pop("KLG_FUSE", None)
    assert np.array_equal(got["stages"], ref["stages"])
    assert np.array_equal(got["per_voice"].view(np.uint32), ref["per_voice"].view(np.uint32))
    peak = float(np.abs(ref["per_voice"]).max())
    assert peak > 0 and float(np.abs(got["mix"] - ref["mix"]).max()) <= 1e-5 * peak * np.sqrt(s.voices) * 4
    assert np.array_equal(got["mix"].view(np.uint32), again["mix"].view(np.uint32))


def test_script_rendered_in_one_call_equals_block_by_block():
    """klg_script_render_device (the whole block loop of an offline render in one call) against klg_script_play_device per block into
    cleared buffers: the same [blocks][2][n] bit for bit — for a bank that takes the single-launch path and one that does not."""
    import torch
    import klang_amd
    for synths in (2, 80):                                   # 256 voices (one launch per block) / 10,240 voices (separate launches)
        N, B = 128, 12
        outs = []
        for one_call in (False, True):
            bank = klang_amd.SynthBank("sub2a", synths=synths, notes=128, max_block=N)
            V = bank.voices
            rng = np.random.default_rng(3)
            script = klang_amd.EventScript(bank, B)
            first = script.add_records(bank.note_records((np.arange(V) // 128).astype(np.int32), rng.integers(40, 90, size=V).astype(np.int32), rng.uniform(0.3, 1.0, size=V).astype(np.float32)))
            v = np.arange(V)
            script.note_on(v % 4, v, first + v)
            script.note_off(5 + v % 3, v)
            script.commit()
            out = torch.full((B, 2, N), 7.0, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()                      # (the library's launches go to the bank's own stream: see above)
            if one_call:
                script.render_device(0, B, out.data_ptr(), N)
            else:
                out.zero_(); torch.cuda.synchronize()
                for b in range(B):
                    script.play_device(b, out[b].data_ptr(), N)
            bank.sync(); torch.cuda.synchronize()
            outs.append(out.cpu().numpy())
            script.close(); bank.close()
        assert np.abs(outs[0]).max() > 0
        assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)), synths


def test_script_span_replayed_as_a_graph_equals_the_launches():
    """klg_script_render_device captures a span's launches as a hipGraph the first time and replays it afterwards: the same span rendered three
    times (capture + two replays, the voices replay the script from block 0 each time) gives the same bits every time, and the same as with
    KLG_GRAPH=0 (plain launches)."""
    import os
    import torch
    import klang_amd
    N, B = 64, 16
    outs = {}
    for mode in ("graph", "plain"):
        if mode == "plain":
            os.environ["KLG_GRAPH"] = "0"
        try:
            bank = klang_amd.SynthBank("sub2a", synths=3, notes=128, max_block=N)
            V = bank.voices
            rng = np.random.default_rng(4)
            script = klang_amd.EventScript(bank, B)
            first = script.add_records(bank.note_records((np.arange(V) // 128).astype(np.int32), rng.integers(40, 90, size=V).astype(np.int32), rng.uniform(0.3, 1.0, size=V).astype(np.float32)))
            v = np.arange(V)
            script.note_on(np.zeros(V, np.int32), v, first + v)
            script.note_off(6 + v % 3, v)
            script.commit()
            out = torch.zeros((B, 2, N), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            res = []
            for _ in range(3):
                script.render_device(0, B, out.data_ptr(), N)
                bank.sync(); torch.cuda.synchronize()
                res.append(out.cpu().numpy().copy())
            outs[mode] = res
            script.close(); bank.close()
        finally:
            os.environ.pop("KLG_GRAPH", None)
    assert np.abs(outs["graph"][0]).max() > 0
    for r in outs["graph"][1:] + outs["plain"]:
        assert np.array_equal(r.view(np.uint32), outs["graph"][0].view(np.uint32))
