"""Noise::process() without the host: the rand() sequence of Generators::{Basic,Fast}::Noise (klang.h:4947-4951, 5357-5366; klang::random(seed) =
srand 236-240) produced ON THE DEVICE — glibc's generator restated with jump-ahead (klang_amd/csrc/klg_rand.hpp, klg_rand_dev.hpp) — against the C library's own
srand() / rand() running in this process: value for value, in the order the reference's Synth / Effect walks its Noise objects, with the sequence handed back and
forth between the C library (host draws) and the device (per-sample draws)."""
import ctypes as C

import numpy as np
import pytest
import torch

import klang_amd

pytestmark = pytest.mark.gpu
LIBC = C.CDLL("libc.so.6")
LIBC.rand.restype = C.c_int
ST_SUSTAIN, ST_OFF = 1, 3


def libc_draws(count):
    return np.fromiter((LIBC.rand() for _ in range(count)), dtype=np.int64, count=count)


def basic_noise(r):                                    # klang.h:4949  rand() * 2.f / (float)RAND_MAX - 1.f   (fp32, in that order)
    return ((r.astype(np.float32) * np.float32(2.0)) / np.float32(2147483647.0) - np.float32(1.0)).astype(np.float32)


def fast_noise(r):                                     # klang.h:5363  15 random bits under a float exponent, - 257
    bits = (((r.astype(np.uint32) & np.uint32(0x7FFF)) << np.uint32(1)) | np.uint32(0x43800000)).astype(np.uint32)
    return (bits.view(np.float32) - np.float32(257.0)).astype(np.float32)


@pytest.mark.parametrize("ranks,per", [(1, 1), (1, 1000), (63, 31), (64, 32), (65, 30), (700, 256), (4097, 62), (70000, 19), (300000, 3)])
def test_device_draws_equal_the_c_librarys(ranks, per):
    L = klang_amd.lib()
    seed = 1000 + ranks + per
    rstride = (ranks + 63) // 64 * 64
    out = torch.full((per, rstride), -1, dtype=torch.int32, device="cuda")     # (before seeding: the HIP runtime's own start-up draws from the C library's generator)
    out2 = torch.zeros((5, 64), dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    LIBC.srand(seed)
    want = libc_draws(ranks * per + 100)
    L.klg_random_seed(seed)
    assert L.klg_rand_fill_device(C.c_void_p(out.data_ptr()), rstride, ranks, per, C.c_void_p(st.cuda_stream)) == 0, L.klg_last_error()
    st.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got[:, :ranks].T.reshape(-1), want[:ranks * per]), "the device's draws differ from rand()'s"
    assert (got[:, ranks:] == -1).all(), "columns beyond the last rank were written"
    # the C library continues where the device stopped ...
    assert L.klg_rand_sync() == 0
    assert np.array_equal(libc_draws(100), want[ranks * per:])
    # ... and the device where the C library stopped (host draws in between), twice in a row without a host draw in between
    LIBC.srand(seed)
    head = libc_draws(7)
    assert np.array_equal(head, want[:7])
    for rep in range(2):
        assert L.klg_rand_fill_device(C.c_void_p(out2.data_ptr()), 64, 3, 5, C.c_void_p(st.cuda_stream)) == 0
        st.synchronize()
        if 7 + 15 * (rep + 1) <= len(want):
            assert np.array_equal(out2.cpu().numpy()[:, :3].T.reshape(-1), want[7 + 15 * rep: 7 + 15 * (rep + 1)])
    assert L.klg_rand_sync() == 0 and L.klg_rand_sync() == 0
    if 37 + 10 <= len(want):
        assert np.array_equal(libc_draws(10), want[37:47])


NOISE_FX = """klgg 1
kind effect 1
ctl 0
op in 0 -1 -1 -1 0
op noise 1 -1 -1 -1 1
op noise 2 -1 -1 -1 0
op add 3 1 2 -1 0
op add 4 0 3 -1 0
ret 4
end
"""


@pytest.mark.parametrize("staged", ["1", "0"])
def test_noise_effect_spans_and_blocks_draw_in_the_references_order(staged, monkeypatch):
    """K effect objects with two Noise generators each: block after block, instance after instance, sample after sample, generator after generator (klang.h:4208-4216
    over the process's objects) — as one launch per block and as spans, against rand() itself."""
    monkeypatch.setenv("KLG_FX_STAGED", staged)
    K, n = 37, 48
    L = klang_amd.lib()
    st = torch.cuda.Stream()
    for spans in ((1, 1, 1), (3, 2), (5,)):
        bank = klang_amd.FxBank(NOISE_FX, K, max_block=n, channels=1)
        seed = 4242 + sum(spans)
        LIBC.srand(seed)
        L.klg_random_seed(seed)
        with torch.cuda.stream(st):
            for blocks in spans:
                x = torch.zeros((blocks, K, 1, n), dtype=torch.float32, device="cuda") + 0.25
                if blocks == 1:
                    bank.process_device(x.data_ptr(), n, st.cuda_stream)
                else:
                    bank.render_device(x.data_ptr(), blocks, n, st.cuda_stream)
                st.synchronize()
                r = libc_draws(blocks * K * n * 2).reshape(blocks, K, n, 2)   # (the C library's copy of the stream, from where the bank took it)
                want = np.float32(0.25) + (fast_noise(r[..., 0]) + basic_noise(r[..., 1]))
                assert np.array_equal(x.cpu().numpy()[:, :, 0, :].view(np.uint32), want.view(np.uint32)), f"staged={staged} spans={spans}"
                assert L.klg_rand_sync() == 0                          # the device hands the stream back: it must stand where the C library's copy stands
        bank.close()


NOISE_NOTE = """klgg 1
ctl 0
op noise 0 -1 -1 -1 0
op noise 1 -1 -1 -1 1
op add 2 0 1 -1 0
ret 2
end
"""


@pytest.mark.parametrize("S,P,density", [(1, 16, 0.5), (40, 32, 0.3), (300, 16, 0.7), (70, 128, 0.05)])
def test_noise_notes_draw_in_the_order_synth_process_walks_them(S, P, density):
    """Sounding notes take their block's values one note after the other (klang.h:4842-4848), whatever slots they sit in: banks of one and of several
    1,024-voice groups, sparse and dense, notes starting and stopping between blocks — without a host round trip, the stream staying on the device."""
    n = 64
    L = klang_amd.lib()
    bank = klang_amd.SynthBank(NOISE_NOTE, synths=S, notes=P, max_block=n)
    V = S * P
    W = bank.state_bytes // 4
    rng = np.random.default_rng(S * 1000 + P)
    seed = 99 + S
    LIBC.srand(seed)
    L.klg_random_seed(seed)
    sounding = np.zeros(V, bool)
    total = 0
    for block in range(6):
        flip = rng.random(V) < (density if block == 0 else 0.08)
        voices = np.nonzero(flip)[0]
        if len(voices):
            sounding[voices] = ~sounding[voices] if block else True
            recs = np.zeros((len(voices), W), np.uint32)
            recs[:, 0] = np.where(sounding[voices], ST_SUSTAIN, ST_OFF)
            bank.voices_upload(voices.tolist(), recs)
        pv, _ = bank.process_voices(n)
        live = np.nonzero(sounding)[0]
        total += len(live)
        # what rand() itself gives those notes, in slot order (the library has not handed the stream back: LIBC's generator is where block - 1 left it,
        # advanced by exactly what the reference would have drawn)
        r = libc_draws(len(live) * n * 2).reshape(len(live), n, 2)
        want = np.zeros((V, n), np.float32)
        want[live] = basic_noise(r[..., 0]) + fast_noise(r[..., 1])
        assert np.array_equal(pv.view(np.uint32), want.view(np.uint32)), f"block {block}: {len(live)} sounding of {V}"
    assert total > 0
    # the device's stream is where the reference's would be: handed back, rand() continues with the value after the last one drawn
    assert L.klg_rand_sync() == 0
    nxt = LIBC.rand()
    LIBC.srand(seed); libc_draws(total * n * 2)
    assert LIBC.rand() == nxt
    bank.close()


@pytest.mark.parametrize("devs", [(0, 0), (0, 0, 0)])
def test_noise_banks_sharded_over_devices_keep_the_one_sequence(devs):
    """A bank sharded inside the library (klg_init with several ids; here the shards share cuda:0): the rand() sequence is handed from shard to shard in instance
    order — what ONE Synth::process loop over all the notes would draw — for notes (7 instances over 2 / 3 shards) and for an effect bank sharded by instance."""
    L = klang_amd.lib()
    S, P, n = 7, 16, 64
    V, rng = S * P, np.random.default_rng(3)
    on = rng.random(V) < 0.6
    def notes():
        bank = klang_amd.SynthBank(NOISE_NOTE, synths=S, notes=P, max_block=n)
        W = bank.state_bytes // 4
        recs = np.zeros((int(on.sum()), W), np.uint32); recs[:, 0] = ST_SUSTAIN
        bank.voices_upload(np.nonzero(on)[0].tolist(), recs)
        L.klg_random_seed(77)
        out = [bank.process_voices(n)[0].copy() for _ in range(3)]
        bank.close()
        return np.stack(out)
    def effect():
        K = 23
        bank = klang_amd.FxBank(NOISE_FX, K, max_block=n, channels=1)
        L.klg_random_seed(78)
        out = []
        for _ in range(3):
            io = np.full((K, 1, n), 0.25, np.float32)
            bank.process(io); out.append(io.copy())
        bank.close()
        return np.stack(out)
    klang_amd.init([0])
    try:
        ref_n, ref_f = notes(), effect()
        LIBC.srand(77)
        r = libc_draws(int(on.sum()) * n * 2).reshape(-1, n, 2)
        assert np.array_equal(ref_n[0][on].view(np.uint32), (basic_noise(r[..., 0]) + fast_noise(r[..., 1])).view(np.uint32))
        klang_amd.init(list(devs))
        got_n, got_f = notes(), effect()
    finally:
        klang_amd.init([0])
    assert np.array_equal(got_n.view(np.uint32), ref_n.view(np.uint32)) and np.array_equal(got_f.view(np.uint32), ref_f.view(np.uint32))
    assert np.abs(ref_f - 0.25).max() > 0.1
