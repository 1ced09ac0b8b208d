#!/usr/bin/env python3
"""tools/pingpong_steady.py [K ...] — klg_fx_pingpong_x per 256-sample block while the control smoothers are still converging (the first blocks
after the dials were set) and once they are stationary (after 300 blocks): the steady state of a plugin whose dials are not being turned."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
import klang_amd
K, N = int(sys.argv[1]), 256
bank = klang_amd.FxBank("pingpong", K, max_block=N)
io = torch.rand((K, 2, N), device="cuda") - 0.5
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
out = {}
for phase, warm in (("converging", 4), ("stationary", 300)):
    for _ in range(warm): bank.process_device(io.data_ptr(), N, ts.cuda_stream)
    torch.cuda.synchronize(); bank.timing_begin()
    for _ in range(40): bank.process_device(io.data_ptr(), N, ts.cuda_stream)
    torch.cuda.synchronize(); n, ms = bank.timing_end()
    out[phase + "_us"] = 1e3 * ms / n
print(json.dumps(out))
''' % ROOT
for K in [int(x) for x in sys.argv[1:]] or [4096, 16384, 65536]:
    o = subprocess.run([sys.executable, "-c", CHILD, str(K)], capture_output=True, text=True)
    try:
        r = json.loads(o.stdout.strip().splitlines()[-1]); r["K"] = K
        for ph in ("converging", "stationary"):
            r[ph + "_frac_of_8TBps"] = K * 256 * 32 / (r[ph + "_us"] * 1e-6) / 8e12
        print(json.dumps(r), flush=True)
    except Exception:
        print("failed", K, o.stderr[-600:])
