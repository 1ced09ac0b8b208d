#!/usr/bin/env python3
"""tools/reverb_modes.py [K ...] — klg_fx_reverb_q at K instances (default 4096) with the early-reflection sums (0) inside the kernel, (1) as a
launch of their own ahead of it, (2) on a second stream beside the previous block's recursive kernel: time per 256-sample block over a stream of
blocks (wall clock and the launch-stream events), one process per mode."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, time, torch
sys.path.insert(0, %r)
import klang_amd
K, N = int(sys.argv[1]), 256
bank = klang_amd.FxBank("reverb", K, max_block=N)
io = torch.rand((K, 2, N), device="cuda") - 0.5
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
for _ in range(20): bank.process_device(io.data_ptr(), N, ts.cuda_stream)
torch.cuda.synchronize(); bank.timing_begin()
t0 = time.perf_counter()
for _ in range(100): bank.process_device(io.data_ptr(), N, ts.cuda_stream)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
n, ms = bank.timing_end()
print(json.dumps({"wall_us_per_block": 1e6 * dt / 100, "events_us_per_block": 1e3 * ms / n}))
''' % ROOT
for K in [int(x) for x in sys.argv[1:]] or [4096]:
    for mode in (0, 1, 2):
        out = subprocess.run([sys.executable, "-c", CHILD, str(K)], env=dict(os.environ, KLG_FX_REVERB_EARLY=str(mode)), capture_output=True, text=True)
        try:
            r = json.loads(out.stdout.strip().splitlines()[-1])
            r.update(K=K, early_mode=mode, alg_TBps=K * 256 * 312 / (r["wall_us_per_block"] * 1e-6) / 1e12, frac_of_8TBps=K * 256 * 312 / (r["wall_us_per_block"] * 1e-6) / 8e12)
            print(json.dumps(r), flush=True)
        except Exception:
            print("failed", K, mode, out.stderr[-800:])
