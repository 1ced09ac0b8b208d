#!/usr/bin/env python3
"""tools/fx_ablate.py — where a block of the PingPong pipeline kernel spends its time: KLG_FX_ABLATE bit masks switch stages off
(1 ring reads, 2 ring writes, 4 io staging, 8 control recurrences, 16 DC filters, 32 full barriers).  One process per mask (the flag is read per launch)."""
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
for _ in range(100): bank.process_device(io.data_ptr(), N, ts.cuda_stream)
torch.cuda.synchronize(); n, ms = bank.timing_end()
print(json.dumps({"kernel_us": 1e3 * ms / n}))
''' % ROOT
for K in (4096, 65536):
    for mask in (0, 1, 2, 3, 4, 7, 8, 16, 24, 31, 32):
        out = subprocess.run([sys.executable, "-c", CHILD, str(K), "pingpong"], env=dict(os.environ, KLG_FX_ABLATE=str(mask)), capture_output=True, text=True)
        try:
            print(json.dumps({"K": K, "ablate": mask, **json.loads(out.stdout.strip().splitlines()[-1])}), flush=True)
        except Exception:
            print("failed", mask, out.stderr[-500:])
