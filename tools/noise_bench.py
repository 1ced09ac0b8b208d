#!/usr/bin/env python3
"""tools/noise_bench.py — what a block of a bank of NOISE notes costs (VERDICT r4 missing #1): V voices of `hiss >> lpf` + `grit` (a Fast::Noise through a
biquad and a Basic::Noise: two rand() draws per voice and sample, klang.h:4947-4951 / 5357-5366), all sounding, 256-sample blocks.
Prints ms per block as a real-time host sees it (one klg_process_device + wait per block) and back to back (blocks queued, one wait at the end).
KLANG_MI355_LIB=<another libklang_mi355.so> runs the same measurement on another build (round 4's library drew on the host: rand() per voice, sample and
generator + a device round trip per block)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import klang_amd  # noqa: E402

PROGRAM = """klgg 1
ctl 0
node 0 lpf
op noise 0 -1 -1 -1 1
op lpf 1 0 -1 0 0
op noise 2 -1 -1 -1 0
op const 3 -1 -1 -1 3dcccccd
op mul 4 2 3 -1 0
op add 5 1 4 -1 0
ret 5
end
"""


def main():
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    N, P = 256, 128
    f32 = np.float32
    bank = klang_amd.SynthBank(PROGRAM, synths=max(1, V // P), notes=min(P, V), max_block=N)
    V = bank.voices
    W = bank.state_bytes // 4
    words = np.zeros((V, W), np.uint32)
    words[:, 0] = 1                                                    # Sustain
    words[:, 1:6] = np.array([0.02, 0.04, 0.02, -1.56, 0.64], f32).view(np.uint32)   # a gentle low-pass (b0 b1 b2 a1 a2)
    for c0 in range(0, V, 1 << 16):
        bank.voices_upload(np.arange(c0, min(V, c0 + (1 << 16)), dtype=np.int32), words[c0:c0 + (1 << 16)])
    klang_amd.lib().klg_random_seed(1)
    mix = torch.zeros((2, N), dtype=torch.float32, device="cuda")
    torch.cuda.set_stream(torch.cuda.Stream())
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        mix.zero_(); bank.process_device(mix.data_ptr(), N, st)
    torch.cuda.synchronize()
    per_block = []
    for _ in range(blocks):
        t0 = time.perf_counter()
        mix.zero_(); bank.process_device(mix.data_ptr(), N, st); torch.cuda.synchronize()
        per_block.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter()
    for _ in range(blocks):
        mix.zero_(); bank.process_device(mix.data_ptr(), N, st)
    torch.cuda.synchronize()
    queued = (time.perf_counter() - t0) * 1e3 / blocks
    bank.timing_begin()
    for _ in range(blocks):
        mix.zero_(); bank.process_device(mix.data_ptr(), N, st)
    torch.cuda.synchronize()
    n, ms = bank.timing_end()
    per_block.sort()
    print(json.dumps(dict(lib=os.path.relpath(klang_amd.LIB_PATH, ROOT), voices=V, block=N, draws_per_block=V * N * 2, ms_per_block_with_wait_median=per_block[len(per_block) // 2],
                          ms_per_block_with_wait_max=per_block[-1], ms_per_block_queued=queued, render_kernel_ms=ms / max(1, n), deadline_ms=N / 48000 * 1e3,
                          finite=bool(torch.isfinite(mix).all()), peak=float(mix.abs().max()))))


if __name__ == "__main__":
    main()
