"""Multi-device banks behind the C-ABI (klg_init with more than one device id): the library shards a synth bank over the
devices — contiguous ranges of synth instances, one shard (state, stream, event queue) per device — routes every event to the owning
shard and combines the shards' [2][n] blocks (one RCCL all-reduce over distinct GPUs; a device-side add when two shards share a GPU).
On a 1-GPU box the list (0, 0) gives two shards on cuda:0: everything but the collective itself is the code a multi-GPU host runs.
With >= 2 GPUs visible the same test also runs over (0, 1) — RCCL."""
import numpy as np
import pytest

from klg_driver import run_scenario_gpu
from scenario_io import Scenario

pytestmark = pytest.mark.gpu


def scenario(patch, synths, notes, blocks):
    s = Scenario(patch=patch, block=256, blocks=blocks, synths=synths, notes=notes, dump=list(range(blocks)))
    rng = np.random.default_rng(11)
    for sy in range(synths):
        for k in range(notes - 1):
            p = int(rng.integers(36, 97))
            s.on(0 if k % 2 else 1, sy, p, float(rng.uniform(0.3, 1.0)), seed=100 * sy + k)
            if k % 3 == 0:
                s.off(2, sy, p)
        s.on(3, sy, 60, 0.9, seed=7)
        s.on(3, sy, 64, 0.9, seed=8)            # one more than there are slots: voice stealing inside the owning shard
    s.sort()
    return s


def device_lists():
    import torch
    lists = [(0, 0), (0, 0, 0)]
    if torch.cuda.device_count() >= 2:
        lists.append((0, 1))
    return lists


@pytest.mark.parametrize("patch,synths,notes", [("sub2a", 5, 16), ("supersaw", 7, 8)])
def test_sharded_bank_equals_single_device_bank(patch, synths, notes):
    """uneven split (5 instances over 2 / 3 shards, 7 over 2 / 3): per-voice outputs and note stages bit for bit, the mix within the
    summation-order bound (the shards' partial sums are added in another order)"""
    import klang_amd
    s = scenario(patch, synths, notes, 5)
    klang_amd.init([0])
    ref = run_scenario_gpu(s)
    try:
        for devs in device_lists():
            klang_amd.init(list(devs))
            got = run_scenario_gpu(s)
            assert np.array_equal(got["stages"], ref["stages"]), devs
            assert np.array_equal(got["per_voice"].view(np.uint32), ref["per_voice"].view(np.uint32)), devs
            peak = float(np.abs(ref["per_voice"]).max())
            assert float(np.abs(got["mix"] - ref["mix"]).max()) <= 1e-5 * peak * np.sqrt(s.voices) * 4, devs
            assert np.abs(got["mix"]).max() > 0
    finally:
        klang_amd.init([0])


def test_sharded_bank_device_entry_and_controls():
    """klg_process_device on a sharded bank adds the GLOBAL block to the caller's device buffer; controls and mix mode reach every shard"""
    import torch
    import klang_amd
    N = 128
    def run(devs):
        klang_amd.init(list(devs))
        bank = klang_amd.SynthBank("supersaw", synths=6, notes=4, max_block=N)
        for sy in range(6):
            bank.set_control(sy, 1, 0.1 * sy)
            bank.random(sy)
            bank.note_on(sy, 50 + sy, 0.8)
        assert abs(bank.get_control(5, 1) - 0.5) < 1e-6
        mix = torch.full((2, N), 0.25, dtype=torch.float32, device="cuda")
        ts = torch.cuda.Stream()
        with torch.cuda.stream(ts):
            bank.process_device(mix.data_ptr(), N, ts.cuda_stream)
            bank.process_device(mix.data_ptr(), N, ts.cuda_stream)
        bank.sync(); torch.cuda.synchronize()
        out = mix.cpu().numpy()
        bank.close()
        return out
    try:
        a, b = run((0,)), run((0, 0))
        assert np.abs(a - 0.25).max() > 0
        assert float(np.abs(a - b).max()) <= 1e-5 * float(np.abs(a).max())
    finally:
        klang_amd.init([0])
