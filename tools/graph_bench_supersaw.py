#!/usr/bin/env python3
"""tools/graph_bench_supersaw.py — the shipped SuperSaw.k as a RECORDED graph patch (tests/golden/supersaw_recorded.klgg: what the
façade records from examples/SuperSaw.k, dumped with KLANG_MI355_DUMP_GRAPH=1) against the hand-written SuperSaw kernel, same
voices; the graph patch with two voices per lane (default) and with one (KLG_GRAPH_X1=1).  One JSON line per variant."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import klang_amd  # noqa: E402

PROGRAM = open(os.path.join(ROOT, "tests", "golden", "supersaw_recorded.klgg")).read()


def to_graph(r):
    """rec::SuperSaw (37 words: flags | 7 x OsmRec(inc offset duty delta) | AdsrRec(8)) -> the recorded program's record (1 + 7*6 + 9 words)."""
    flags = int(r[0])
    g = np.zeros(52, np.uint32)
    g[0] = flags & 3
    for k in range(7):
        g[1 + 6 * k:5 + 6 * k] = r[1 + 4 * k:5 + 4 * k]
        g[5 + 6 * k] = (flags >> (8 + 2 * k)) & 3
    a = 1 + 4 * 7
    g[43:47] = r[a:a + 4]; g[47] = (flags >> 2) & 0x3F; g[48:52] = r[a + 4:a + 8]
    return g


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
    return dict(value=bank.voices * N * steps / dt, kernel_ms=ms / n)


def main():
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
    N, P = 256, 32
    rng = np.random.default_rng(7)
    pitches = rng.integers(36, 97, size=V)
    hand = klang_amd.SynthBank("supersaw", synths=V // P, notes=P, max_block=N)
    hand.note_on_many(np.arange(V) // P, pitches, np.full(V, 0.8, np.float32))
    mix = np.zeros((2, N), np.float32); hand.process(mix)                     # events applied
    words = np.stack([to_graph(hand.voice_download(v)) for v in range(4096)])   # 4096 distinct voices, tiled over the bank
    out = [dict(timed(hand, N, 50, 30), variant="hand-written (one voice per lane)", voices=V)]
    print(json.dumps(out[-1]), flush=True)
    hand.close()
    for variant, env in (("recorded graph, two voices per lane", None), ("recorded graph, one voice per lane", "1")):
        if env:
            os.environ["KLG_GRAPH_X1"] = env
        bank = klang_amd.SynthBank(PROGRAM, synths=V // P, notes=P, max_block=N)
        os.environ.pop("KLG_GRAPH_X1", None)
        assert bank.state_bytes == 52 * 4
        for c0 in range(0, V, 4096):
            bank.voices_upload(np.arange(c0, c0 + 4096, dtype=np.int32), words)
        out.append(dict(timed(bank, N, 50, 30), variant=variant, voices=V))
        print(json.dumps(out[-1]), flush=True)
        bank.close()


if __name__ == "__main__":
    main()
