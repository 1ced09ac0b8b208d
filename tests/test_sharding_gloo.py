"""The N > 1 path on CPU: world_size-2 `gloo` processes exercise klang_amd.shard (voice partition, event routing,
one all-reduce of the stereo block per step).  The per-rank renderer is the TEST-ONLY oracle injected through
`bank_factory` (no GPU here); the product default is the HIP SynthBank."""
import ctypes as C
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import ctypes as C, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from klang_amd.shard import ShardedSynthBank, shard_range, owner_of

ko = C.CDLL(os.path.join(ROOT, "oracle", "_build", "libklang_oracle.so"))
ko.ko_bank_create.restype = C.c_void_p
ko.ko_bank_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float]
ko.ko_bank_note_on.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_long]
ko.ko_bank_note_off.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float]
ko.ko_bank_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
ko.ko_patch_from_name.argtypes = [C.c_char_p]

class OracleBank:                      # CPU stand-in with SynthBank's interface (checker only)
    def __init__(self, patch, synths, notes, fs, max_block):
        self.h = ko.ko_bank_create(ko.ko_patch_from_name(patch.encode()), synths, notes, C.c_float(fs))
        self.voices = synths * notes
    def random(self, seed): pass
    def note_on(self, s, p, v): return ko.ko_bank_note_on(self.h, s, p, C.c_float(v), -1)
    def note_off(self, s, p, v): ko.ko_bank_note_off(self.h, s, p, C.c_float(v))
    def set_control(self, s, i, v): pass
    def process_device(self, ptr, n, stream=None): ko.ko_bank_process(self.h, None, C.c_void_p(ptr), None, n)
    def close(self): pass

rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
SYNTHS, NOTES, N, B = 5, 16, 128, 6           # 5 instances over 2 ranks: uneven split 3 + 2
rng = np.random.default_rng(42)
events = [(int(rng.integers(0, 3)), int(rng.integers(0, SYNTHS)), int(rng.integers(36, 97)), float(rng.uniform(.3, 1))) for _ in range(60)]
offs = [(3, s, p) for (_, s, p, _) in events[::3]]
bank = ShardedSynthBank("sub2a", SYNTHS, NOTES, max_block=N, rank=rank, world=world, bank_factory=OracleBank)
single = OracleBank("sub2a", SYNTHS, NOTES, 48000.0, N) if rank == 0 else None
assert (bank.lo, bank.hi) == shard_range(SYNTHS, world, rank)
assert all(owner_of(s, SYNTHS, world) == (0 if s < 3 else 1) for s in range(SYNTHS))
worst = 0.0
for b in range(B):
    for (eb, s, p, v) in events:
        if eb == b:
            bank.note_on(s, p, v)
            if single: single.note_on(s, p, v)
    for (ob, s, p) in offs:
        if ob == b:
            bank.note_off(s, p)
            if single: single.note_off(s, p, 0.0)
    mix = torch.zeros((2, N), dtype=torch.float32)
    bank.process_device(mix, N)
    if single:
        ref = torch.zeros((2, N), dtype=torch.float32)
        single.process_device(ref.data_ptr(), N)
        peak = max(float(ref.abs().max()), 1e-9)
        worst = max(worst, float((mix - ref).abs().max()) / peak)
# every rank must hold the same reduced block
g = [torch.zeros_like(mix) for _ in range(world)]
dist.all_gather(g, mix)
assert all(torch.equal(g[0], x) for x in g)
if rank == 0:
    print("WORST", worst)
    assert worst < 1e-5, worst
dist.destroy_process_group()
'''


def test_shard_range_partition():
    from klang_amd.shard import owner_of, shard_range
    for n in (1, 7, 8, 1000, 8192):
        for world in (1, 2, 3, 8):
            if n < world:
                continue
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
            for item in (0, n // 2, n - 1):
                r = owner_of(item, n, world)
                assert spans[r][0] <= item < spans[r][1]


def test_two_rank_gloo_mix_equals_single_process(oracle_build, tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {ROOT!r}\n" + WORKER)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(port)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    assert "WORST" in outs[0][0]


FX_WORKER = r'''
import ctypes as C, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, ROOT)
from klang_amd.shard import ShardedFxBank, shard_range

ko = C.CDLL(os.path.join(ROOT, "oracle", "_build", "libklang_oracle.so"))
ko.ko_fxbank_create.restype = C.c_void_p
ko.ko_fxbank_create.argtypes = [C.c_int, C.c_int, C.c_float]
ko.ko_fxbank_control.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float]
ko.ko_fxbank_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
ko.ko_patch_from_name.argtypes = [C.c_char_p]

class OracleFx:                        # CPU stand-in with FxBank's interface (checker only)
    def __init__(self, patch, instances, fs=48000.0, max_block=256):
        self.h = ko.ko_fxbank_create(ko.ko_patch_from_name(patch.encode()), instances, C.c_float(fs)); assert self.h
        self.instances = instances
    def set_control(self, k, i, v): ko.ko_fxbank_control(self.h, k, i, C.c_float(v))
    def process(self, io):
        assert io.dtype == np.float32 and io.flags.c_contiguous and io.shape[0] == self.instances
        ko.ko_fxbank_process(self.h, io.ctypes.data_as(C.c_void_p), io.shape[2]); return io
    def close(self): pass

rank, world, port, patch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
K, N, B = 5, 96, 5                          # 5 instances over 2 ranks: 3 + 2
bank = ShardedFxBank(patch, K, max_block=N, rank=rank, world=world, bank_factory=OracleFx)
whole = OracleFx(patch, K, 48000.0, N)      # every rank also runs the unsharded bank: the check needs no exchange
assert (bank.lo, bank.hi) == shard_range(K, world, rank)
rng = np.random.default_rng(7)
dials = [(int(rng.integers(0, B)), int(rng.integers(0, K)), int(rng.integers(0, 3)), float(rng.uniform(.1, .9))) for _ in range(12)]
mine = 0
for b in range(B):
    for (db, k, i, v) in dials:
        if db == b:
            bank.set_control(k, i, v); whole.set_control(k, i, v); mine += bank.owns(k)
    io = (rng.random((K, 2, N), dtype=np.float32) - 0.5).astype(np.float32)
    ref = whole.process(io.copy())
    loc = np.ascontiguousarray(bank.local(io))
    bank.process(loc)
    assert np.array_equal(loc.view(np.uint32), ref[bank.lo:bank.hi].view(np.uint32)), (rank, b)      # bit-exact, row for row
# the ranks' rows together are the whole block (gathered here only to check it; the data path itself has no collective)
parts = [None] * world
dist.all_gather_object(parts, (bank.lo, bank.hi, loc))
got = np.concatenate([p[2] for p in sorted(parts, key=lambda p: p[0])])
assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
counts = [None] * world
dist.all_gather_object(counts, mine)
assert sum(counts) == len(dials)            # every dial reached exactly one rank
if rank == 0: print("FXOK", patch)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("patch", ["pingpong", "reverb"])
def test_two_rank_gloo_effect_instances_shard_without_a_collective(oracle_build, tmp_path, patch):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "fx_worker.py"
    script.write_text(f"ROOT = {ROOT!r}\n" + FX_WORKER)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(port), patch], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    assert "FXOK" in outs[0][0]


def test_sharded_fx_bank_refuses_an_empty_rank():
    from klang_amd.shard import ShardedFxBank
    with pytest.raises(ValueError):
        ShardedFxBank("pingpong", 1, rank=1, world=2, bank_factory=lambda *a, **k: None)
