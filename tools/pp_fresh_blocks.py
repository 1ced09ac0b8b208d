#!/usr/bin/env python3
"""tools/pp_fresh_blocks.py [instances] — a FRESH PingPong bank handed over one 256-sample block per call (a real-time host), bench.py's block-by-block leg taken
apart: kernel time per block over the first 75 blocks (the dial smoothers of a new object still converge: the general forms) and over the other 300 (dials at
rest), and the average the leg reports.  KLANG_MI355_LIB / KLG_FX_PINGPONG_MV select the build / the launch plan under measurement."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, klang_amd
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N, BLOCKS, HEAD = 256, 375, 75
bank = klang_amd.FxBank("pingpong", K, max_block=N)
g = torch.Generator(device="cuda").manual_seed(1)
io = torch.zeros((1, K, 2, N), device="cuda")
burst = torch.rand((19, K, 2, N), device="cuda", generator=g) - 0.5
ts = torch.cuda.Stream()
with torch.cuda.stream(ts):
    st = ts.cuda_stream
    bank.render_device(io.data_ptr(), 1, N, st); torch.cuda.synchronize()
    out = {}
    bank.timing_begin()
    for b in range(BLOCKS):
        if b == HEAD:
            out["head_us_per_block"] = 1e3 * bank.timing_end()[1] / HEAD; bank.timing_begin()
        io.zero_()
        if b < 19: io[0].copy_(burst[b])
        bank.process_device(io.data_ptr(), N, st)
    torch.cuda.synchronize()
    out["rest_us_per_block"] = 1e3 * bank.timing_end()[1] / (BLOCKS - HEAD)
out["leg_us_per_block"] = (out["head_us_per_block"] * HEAD + out["rest_us_per_block"] * (BLOCKS - HEAD)) / BLOCKS
out["frac_of_hbm_peak"] = K * N * 32 / (out["leg_us_per_block"] * 1e-6) / 8e12
out["checksum"] = float(io.double().abs().sum().item())
print(json.dumps({k: round(v, 4) if isinstance(v, float) else v for k, v in out.items()}))
