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


@pytest.mark.parametrize("patch,synths,notes", [("sub2a", 5, 16), ("supersaw", 7, 8), ("fm4", 5, 8), ("fm3", 3, 4)])
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


def test_fm4_share_of_config_5_over_two_shards(oracle_build):
    """BASELINE config 5's shape on one box: 131,072 FM4 voices as TWO shards of 65,536 (klg_init([0, 0]); two GPUs: (0, 1) and RCCL) —
    256 class instances against the oracle, every replica in either shard bit-exact with its class, the combined mix = the fp64 sum of all
    voices (voices shard, one reduce of the [2][n] block)."""
    import torch
    import klang_amd
    from klg_driver import rel_err, run_scenario_oracle
    from test_gpu_fullsize import class_scenario, replicate
    classes, notes, total = 128, 32, 4096                                 # 4,096 instances x 32 notes = 131,072 voices; 2,048 instances per shard
    dump = [0, 2, 4]
    small = class_scenario("fm4", classes, notes, 5, 256, 4242, False, dump)
    ref = run_scenario_oracle(small, oracle_build)
    big = replicate(small, total)
    devs = [0, 1] if torch.cuda.device_count() >= 2 else [0, 0]
    klang_amd.init(devs)
    try:
        got = run_scenario_gpu(big)
    finally:
        klang_amd.init([0])
    pv = got["per_voice"]
    Vc = small.voices
    assert pv.shape[1] == total * notes
    assert rel_err(pv[:, :Vc], ref["per_voice"]) <= 1e-5
    assert np.array_equal(got["stages"][:, :Vc], ref["stages"])
    reps = pv.reshape(len(dump), total // classes, Vc, -1)
    assert np.array_equal(reps.view(np.uint32), np.broadcast_to(reps[:, :1], reps.shape).view(np.uint32)), "a replica (in either shard) differs from its class representative"
    V = pv.shape[1]
    for i, b in enumerate(dump):
        want = pv[i].astype(np.float64).sum(axis=0)
        peak = float(np.abs(pv[i]).max())
        assert float(np.abs(got["mix"][b, 0] - want).max()) <= 4 * np.sqrt(V) * np.finfo(np.float32).eps * peak * np.sqrt(V)
    assert float(np.abs(got["mix"]).max()) > 0


def test_config_5_at_its_literal_size_over_eight_shards(oracle_build):
    """BASELINE config 5 as written — 1,048,576 FM4 voices over 8 GPUs, one reduce of the stereo block — on whatever this box has: klg_init with EIGHT ids
    (the GPUs there are, repeated: eight shares of 131,072 voices each; a device listed twice combines by a plain add, distinct devices by the one
    ncclAllReduce).  256 class instances against the oracle, every one of the 128 replicas of a class — they lie in all eight shards — bit-exact with its
    class, note stages equal, the combined mix = the fp64 sum of all voices.  The first 8-GPU run must not be the first 8-way execution."""
    import torch
    import klang_amd
    from klg_driver import rel_err, run_scenario_oracle
    from test_gpu_fullsize import class_scenario, replicate
    classes, notes, total = 256, 32, 32768                                # 32,768 instances x 32 notes = 1,048,576 voices; 4,096 instances = 131,072 voices per shard
    dump = [0, 3]
    small = class_scenario("fm4", classes, notes, 5, 256, 777, False, dump)
    ref = run_scenario_oracle(small, oracle_build)
    big = replicate(small, total)
    have = max(1, torch.cuda.device_count())
    devs = [i % have for i in range(8)]
    klang_amd.init(devs)
    try:
        got = run_scenario_gpu(big)
    finally:
        klang_amd.init([0])
    pv = got["per_voice"]
    Vc = small.voices
    assert pv.shape[1] == total * notes == 1 << 20
    assert rel_err(pv[:, :Vc], ref["per_voice"]) <= 1e-5
    assert np.array_equal(got["stages"][:, :Vc], ref["stages"])
    reps = pv.reshape(len(dump), total // classes, Vc, -1)
    for r in range(1, total // classes):                                  # (replica r lies in shard r * 8 // 128)
        assert np.array_equal(reps[:, r].view(np.uint32), reps[:, 0].view(np.uint32)), f"replica {r} (shard {r * 8 // (total // classes)}) differs from its class representative"
    st = got["stages"].reshape(got["stages"].shape[0], total // classes, Vc)
    assert (st == st[:, :1]).all(), "note stages differ between replicas"
    V = pv.shape[1]
    for i, b in enumerate(dump):
        want = pv[i].astype(np.float64).sum(axis=0)
        peak = float(np.abs(pv[i]).max())
        assert float(np.abs(got["mix"][b, 0] - want).max()) <= 4 * np.sqrt(V) * np.finfo(np.float32).eps * peak * np.sqrt(V)
    assert float(np.abs(got["mix"]).max()) > 0


@pytest.mark.parametrize("patch", ["pingpong", "reverb"])
def test_effect_bank_is_sharded_by_instance(patch):
    """effects shard BY INSTANCE, no collective.  Under klg_init with several ids an effect bank is dealt to the devices in
    contiguous ranges (uneven: 70 instances over 2 / 3 shards); per-instance controls reach the owning shard; the result equals the
    single-device bank bit for bit; the device entry refuses loudly (a device block lives on one GPU)."""
    import klang_amd
    K, N, B = (70, 128, 6) if patch == "pingpong" else (70, 256, 14)      # (Reverb.k's first early reflection arrives after 50 ms = 2,400 samples)
    rng = np.random.default_rng(3)
    x = rng.uniform(-0.5, 0.5, size=(B, K, 2, N)).astype(np.float32)
    x[3:] = 0
    ctl = [(k, c, float(rng.uniform(0.05, 0.9))) for k in range(K) for c in ((0, 1, 5) if patch == "pingpong" else (2, 3, 6, 7))]

    def run(devs):
        klang_amd.init(list(devs))
        bank = klang_amd.FxBank(patch, K, max_block=N)
        for k, c, v in ctl:
            bank.set_control(k, c, v)
        out = []
        for b in range(B):
            if b == 2:
                bank.set_control(K - 1, ctl[-1][1], 0.33)                  # a change mid-run, last instance (the last shard)
            io = x[b].copy(); bank.process(io); out.append(io)
        back = [bank.get_control(k, 1) for k in (0, K // 2, K - 1)]
        recs = [bank.download_record(k) for k in (0, K // 2, K - 1)]              # an instance's record comes from the shard that owns it
        bank.upload_words(K - 1, 0, recs[-1][:4])                                # ... and uploads go there (the same words back: nothing changes)
        assert bank.record_words() == recs[0].size and np.array_equal(bank.download_record(K - 1), recs[-1])
        if len(devs) > 1:
            import torch
            with pytest.raises(klang_amd.KlangError, match="sharded over"):
                bank.process_device(torch.zeros((K, 2, N), device="cuda").data_ptr(), N)
        bank.close()
        return np.stack(out), (back, [r.tobytes() for r in recs])

    try:
        ref, ref_back = run((0,))
        for devs in device_lists():
            got, back = run(devs)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), devs
            assert back == ref_back
        assert np.abs(ref[-1]).max() > 1e-4                                   # the input stopped after block 2: what sounds now came out of the rings
    finally:
        klang_amd.init([0])


def test_two_banks_driven_from_two_host_threads():
    """One calling thread per handle is the contract (include/klang_mi355.h); two handles may be driven CONCURRENTLY.  The bank's device is
    a field of the handle and is bound per call and per thread (no process-global device is swapped around a call): two threads, each
    creating, playing and reading its own bank — one of them a sharded (0, 0) bank —, get what they get alone."""
    import threading
    import klang_amd
    s = scenario("sub2a", 4, 8, 6)

    def play(result, key):
        result[key] = run_scenario_gpu(s)

    klang_amd.init([0])
    alone = run_scenario_gpu(s)
    try:
        klang_amd.init([0, 0])
        sharded_alone = run_scenario_gpu(s)
        res = {}
        for rnd in range(3):
            threads = [threading.Thread(target=play, args=(res, i)) for i in range(2)]
            for t in threads: t.start()
            for t in threads: t.join()
            for i in range(2):
                assert np.array_equal(res[i]["per_voice"].view(np.uint32), alone["per_voice"].view(np.uint32)), (rnd, i)
                assert np.array_equal(res[i]["mix"].view(np.uint32), sharded_alone["mix"].view(np.uint32)), (rnd, i)
    finally:
        klang_amd.init([0])
