#!/usr/bin/env python3
"""tools/fx_ablate.py [masks...] — kernel time per 256-sample block of the two effect kernels at several bank sizes (HIP events on the launch
stream).  With masks: KLG_FX_ABLATE values for the ONE-WAVE PingPong kernel (1 ring reads, 2 ring writes, 4 io staging off), one process each."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, time, torch
sys.path.insert(0, %r)
import klang_amd
K, N = int(sys.argv[1]), 256
bank = klang_amd.FxBank(sys.argv[2], K, max_block=N)
io = torch.rand((K, 2, N), device="cuda") - 0.5
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
for _ in range(10): bank.process_device(io.data_ptr(), N, ts.cuda_stream)
torch.cuda.synchronize(); bank.timing_begin()
for _ in range(60): bank.process_device(io.data_ptr(), N, ts.cuda_stream)
torch.cuda.synchronize(); n, ms = bank.timing_end()
print(json.dumps({"kernel_us": 1e3 * ms / n}))
''' % ROOT
masks = [int(m) for m in sys.argv[1:]] or [0]
BYTES = {"pingpong": 32, "reverb": 312}
for patch, sizes in (("pingpong", (4096, 16384, 65536)), ("reverb", (1024, 4096, 6144, 8192, 12288, 16384))):
    for K in sizes:
        for mask in (masks if patch == "pingpong" else [0]):
            out = subprocess.run([sys.executable, "-c", CHILD, str(K), patch], env=dict(os.environ, KLG_FX_ABLATE=str(mask)), capture_output=True, text=True)
            try:
                r = json.loads(out.stdout.strip().splitlines()[-1])
                r.update(patch=patch, K=K, ablate=mask, alg_TBps=K * 256 * BYTES[patch] / (r["kernel_us"] * 1e-6) / 1e12)
                print(json.dumps(r), flush=True)
            except Exception:
                print("failed", patch, K, mask, out.stderr[-800:])
