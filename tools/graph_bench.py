#!/usr/bin/env python3
"""tools/graph_bench.py — recorded-graph patch vs hand-written kernel for the same patch (sub2a), same voices, one MI355X.
The graph program is what the DSL facade records from tests/patches/sub2a.k; voice records are converted from the
hand-written bank's records, so both banks hold identical state.  Prints one JSON line per variant."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import klang_amd  # noqa: E402
from test_gpu_graph import SUB2A_PROGRAM, sub2a_to_graph  # noqa: E402


def timed(bank, N, steps, warmup):
    mix = torch.zeros((2, N), dtype=torch.float32, device="cuda")
    torch.cuda.set_stream(torch.cuda.Stream())      # a stream of our own: handle 0 (torch's default) means "the bank's own stream" to the library, which torch's clears are not ordered with
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(warmup):
        mix.zero_(); bank.process_device(mix.data_ptr(), N, st)
    torch.cuda.synchronize(); bank.timing_begin(); t0 = time.perf_counter()
    for _ in range(steps):
        mix.zero_(); bank.process_device(mix.data_ptr(), N, st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n, ms = bank.timing_end()
    return dict(value=bank.voices * N * steps / dt, kernel_ms=ms / n, mix_abs_sum=float(mix.abs().sum().item()))


def main():
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    N, P = 256, 128
    rng = np.random.default_rng(20250314)
    pitches = rng.integers(36, 97, size=V)
    # one record per pitch from the hand-written bank's own on() code
    proto = klang_amd.SynthBank("sub2a", synths=1, notes=61, max_block=N)
    for p in range(36, 97):
        proto.note_on(0, p, 0.8)
    recs = {36 + i: sub2a_to_graph(proto.voice_download(i)) for i in range(61)}
    proto.close()
    out = []
    for variant in ("hand_x2", "hand_x1", "graph"):
        if variant == "graph":
            bank = klang_amd.SynthBank(SUB2A_PROGRAM, synths=V // P, notes=P, max_block=N)
            words = np.stack([recs[int(p)] for p in pitches])
            for c0 in range(0, V, 1 << 16):
                bank.voices_upload(np.arange(c0, min(V, c0 + (1 << 16)), dtype=np.int32), words[c0:c0 + (1 << 16)])
        else:
            if variant == "hand_x1":
                os.environ["KLG_RENDER_X1"] = "1"
            bank = klang_amd.SynthBank("sub2a", synths=V // P, notes=P, max_block=N)
            os.environ.pop("KLG_RENDER_X1", None)
            for v in range(V):
                bank.note_on(v // P, int(pitches[v]), 0.8)
        r = timed(bank, N, 100, 30)          # 30 warmup blocks: past attack/decay, every voice holds at sustain
        r.update(variant=variant, voices=V, block=N, state_bytes=bank.state_bytes)
        print(json.dumps(r), flush=True)
        out.append(r)
        bank.close()
    print(json.dumps({"graph_vs_hand_x1": out[2]["kernel_ms"] / out[1]["kernel_ms"], "graph_vs_hand_x2": out[2]["kernel_ms"] / out[0]["kernel_ms"]}))


if __name__ == "__main__":
    main()
