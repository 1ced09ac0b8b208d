#!/usr/bin/env python3
"""tools/phase_run.py <sustain|ramp> [voices] — the Subtractive bank in ONE envelope phase for counter passes (tools/pmc_any.py averages the
later half of the launches): `sustain` = every voice holding at its sustain point (the wave-uniform short loop), `ramp` = every voice inside
its release ramp (note-off for all, then blocks within the 0.25 s release)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import klang_amd  # noqa: E402

phase = sys.argv[1]
V = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
N = 256
bank = klang_amd.SynthBank("sub2a", synths=V // 128, notes=128, max_block=N)
bank.random(1)
rng = np.random.default_rng(3)
pitches = rng.integers(36, 97, size=V).astype(np.int32)
synth = (np.arange(V) // 128).astype(np.int32)
bank.note_on_many(synth, pitches, np.full(V, 0.8, np.float32))
mix = torch.zeros((2, N), dtype=torch.float32, device="cuda")
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
for _ in range(30):                                   # attack + decay: 1.25 ms + 0.25 s... every voice at its sustain point
    mix.zero_(); bank.process_device(mix.data_ptr(), N, st)
if phase == "ramp":
    bank.note_off_many(synth, pitches, np.zeros(V, np.float32))
    blocks = 40                                       # the release lasts 48 blocks
else:
    blocks = 60
for _ in range(blocks):
    mix.zero_(); bank.process_device(mix.data_ptr(), N, st)
torch.cuda.synchronize()
print(phase, V, float(mix.abs().sum().item()))
bank.close()
