#!/usr/bin/env python3
"""tools/recorded_small_banks.py [forms] — RECORDED note programs at plug-in sizes: tests/patches/sub2a.k, examples/FM.k and examples/SuperSaw.k as the facade records
them, 32 .. 16,384 voices all sounding (records converted from the hand-written banks' own on() code: both hold the same state), kernel time per 256-sample block in
the forms named (KLG_GRAPH_SP: 0 = a voice per lane / two voices per lane, 1 = a voice per wave, 8 = eight voices per wave, default = what the library picks) next to
the hand-written kernel of the same patch.  One JSON line per (patch, voices)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, klang_amd  # noqa: E402
from test_gpu_graph import SUB2A_PROGRAM, sub2a_to_graph  # noqa: E402
from graph_bench_supersaw import PROGRAM as SUPERSAW_PROGRAM, to_graph as supersaw_to_graph  # noqa: E402
FM_PROGRAM = open(os.path.join(ROOT, "tests", "golden", "fm_recorded.klgg")).read()


def fm_to_graph(r):
    """rec::FM<3> (flags, meta, 3 x OpRec(inc pos r_out r_target r_rate time px[2] py[2]), AdsrRec(8)) -> the recorded FM.k's record: 3 x operator
    (inc pos frequency amp | r_out r_target r_rate time bits npoints loop px[4] py[4] + 12 + 12 more point slots) + adsr(9)"""
    flags, meta = int(r[0]), int(r[1])
    OPW = 4 + 15 + 24
    g = np.zeros(1 + 3 * OPW + 9, np.uint32)
    g[0] = flags & 3
    for k in range(3):
        o = r[2 + 10 * k:12 + 10 * k]; b = 1 + OPW * k
        g[b] = o[0]; g[b + 1] = o[1]; g[b + 2] = 0; g[b + 3] = np.float32(1.0).view(np.uint32)
        g[b + 4:b + 8] = o[2:6]
        g[b + 8] = (flags >> (8 + 6 * k)) & 0x3F
        g[b + 9] = (meta >> (2 * k)) & 3
        g[b + 10] = 0xFFFF
        g[b + 11:b + 13] = o[6:8]; g[b + 15:b + 17] = o[8:10]
    a = 2 + 30; b = 1 + 3 * OPW
    g[b:b + 4] = r[a:a + 4]; g[b + 4] = (flags >> 2) & 0x3F; g[b + 5:b + 9] = r[a + 4:a + 8]
    return g


CASES = {"sub2a": ("sub2a", SUB2A_PROGRAM, sub2a_to_graph, 128), "fm3": ("fm3", FM_PROGRAM, fm_to_graph, 32), "supersaw": ("supersaw", SUPERSAW_PROGRAM, supersaw_to_graph, 32)}


def timed(bank, N=256, steps=100, warmup=40):
    mix = torch.zeros((2, N), dtype=torch.float32, device="cuda")
    torch.cuda.set_stream(torch.cuda.Stream())
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(warmup):
        mix.zero_(); bank.process_device(mix.data_ptr(), N, st)
    torch.cuda.synchronize(); bank.timing_begin(); t0 = time.perf_counter()
    for _ in range(steps):
        mix.zero_(); bank.process_device(mix.data_ptr(), N, st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n, ms = bank.timing_end()
    return round(1e3 * ms / n, 2), round(1e6 * dt / steps, 2), float(mix.abs().sum().item())


def main():
    forms = sys.argv[1:] or ["default", "0"]
    rng = np.random.default_rng(7)
    for name, (patch, program, conv, P) in CASES.items():
        for V in [int(x) for x in os.environ.get("RSB_VOICES", "32,1024,4096,16384").split(",")]:
            P_ = min(P, V)
            pitches = rng.integers(36, 97, size=V)
            hand = klang_amd.SynthBank(patch, synths=V // P_, notes=P_, max_block=256)
            hand.note_on_many(np.arange(V) // P_, pitches, np.full(V, 0.8, np.float32))
            hand.process(np.zeros((2, 256), np.float32))
            words = np.stack([conv(hand.voice_download(v)) for v in range(min(V, 1024))])
            row = {"patch": name, "voices": V, "hand_kernel_us": timed(hand)[0]}
            hand.close()
            for form in forms:
                if form != "default":
                    os.environ["KLG_GRAPH_SP"] = form
                bank = klang_amd.SynthBank(program, synths=V // P_, notes=P_, max_block=256)
                os.environ.pop("KLG_GRAPH_SP", None)
                assert bank.state_bytes == words.shape[1] * 4, (bank.state_bytes, words.shape)
                for c0 in range(0, V, 1024):
                    bank.voices_upload(np.arange(c0, min(V, c0 + 1024), dtype=np.int32), words[:min(1024, V - c0)])
                k, s, chk = timed(bank)
                row[f"recorded_sp_{form}_kernel_us"] = k; row[f"recorded_sp_{form}_step_us"] = s
                bank.close()
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
