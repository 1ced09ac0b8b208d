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
    script.play_device(0, mix.data_ptr(), 64)
    bank.sync()
    assert float(mix.abs().sum()) > 0
    bank.close()                                          # the bank goes first
    with pytest.raises(klang_amd.KlangError, match="destroyed"):
        script.play_device(1, mix.data_ptr(), 64)
    script.close()                                        # and the handle can still be released
