#!/usr/bin/env python3
"""tools/graph_width_bench.py — one or two voices per lane?  A generated patch of k duty-0 saws summed through a biquad and an ADSR
(k = 1 is the Subtractive patch), rendered both ways (KLG_GRAPH_X1=1 / KLG_GRAPH_X2=1 force the width), with the register count
klg_synth_create_graph bases its choice on (KLG_GRAPH_DEBUG).  The data behind the ~160-register rule."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import klang_amd  # noqa: E402


def program(k, duty):
    lines = ["klgg 1", "ctl 0"] + [f"node {i} saw" for i in range(k)] + [f"node {k} lpf", f"node {k + 1} adsr"]
    r = 0
    acc = None
    for i in range(k):
        lines.append(f"op osc {r} -1 -1 {i} 0"); o = r; r += 1
        if acc is None:
            acc = o
        else:
            lines.append(f"op add {r} {acc} {o} -1 0"); acc = r; r += 1
    lines.append(f"op lpf {r} {acc} -1 {k} 0"); f = r; r += 1
    lines.append(f"op env {r} -1 -1 {k + 1} 0"); e = r; r += 1
    lines.append(f"op mul {r} {f} {e} -1 0"); m = r
    lines += [f"op stopif -1 -1 -1 {k + 1} 0", f"ret {m}", "end"]
    return "\n".join(lines) + "\n"


def run(k, duty, width, V=1 << 19, N=256):
    os.environ.pop("KLG_GRAPH_X1", None); os.environ.pop("KLG_GRAPH_X2", None)
    os.environ["KLG_GRAPH_X1" if width == 1 else "KLG_GRAPH_X2"] = "1"
    bank = klang_amd.SynthBank(program(k, duty), synths=V // 128, notes=128, max_block=N)
    W = bank.state_bytes // 4
    f32 = np.float32
    rng = np.random.default_rng(1)
    words = np.zeros((V, W), np.uint32)
    words[:, 0] = 1
    for i in range(k):
        inc = rng.integers(1 << 20, 1 << 26, V).astype(np.uint32)
        words[:, 1 + 6 * i] = inc; words[:, 2 + 6 * i] = rng.integers(0, 1 << 32, V, dtype=np.uint64).astype(np.uint32)
        words[:, 3 + 6 * i] = (1 << 29) if duty else 0
        words[:, 4 + 6 * i] = (((inc >> 9) | 0x3f800000).astype(np.uint32).view(f32) - f32(1)).view(np.uint32)
    b = 1 + 6 * k
    words[:, b:b + 5] = np.array([0.1, 0.2, 0.1, -0.5, 0.2], f32).view(np.uint32)
    a = b + 9
    words[:, a] = f32(0.7).view(np.uint32); words[:, a + 1] = f32(0.7).view(np.uint32); words[:, a + 4] = 2 << 2; words[:, a + 7] = f32(0.7).view(np.uint32); words[:, a + 8] = f32(0.3).view(np.uint32)
    for c0 in range(0, V, 1 << 16):
        bank.voices_upload(np.arange(c0, c0 + (1 << 16), dtype=np.int32), words[c0:c0 + (1 << 16)])
    mix = torch.zeros((2, N), dtype=torch.float32, device="cuda"); torch.cuda.set_stream(torch.cuda.Stream()); st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        mix.zero_(); bank.process_device(mix.data_ptr(), N, st)
    torch.cuda.synchronize(); bank.timing_begin()
    for _ in range(30):
        mix.zero_(); bank.process_device(mix.data_ptr(), N, st)
    torch.cuda.synchronize(); n, ms = bank.timing_end()
    lanes = bank.voices_per_lane
    bank.close()
    return dict(saws=k, duty=bool(duty), forced_voices_per_lane=width, ran_voices_per_lane=lanes, kernel_ms=ms / n, value=V * N / (ms / n * 1e-3))


if __name__ == "__main__":
    for duty in (0, 1):
        for k in (1, 2, 3, 4, 5, 7):
            for width in (1, 2):
                print(json.dumps(run(k, duty, width)), flush=True)
