#!/usr/bin/env python3
"""tools/pp_span_variants.py — why the bench leg's steady PingPong spans (11.2 us per block) are slower than tools/fx_span_bench.py's (10.35): the same
bank driven with the differences one at a time (span length, silence in, a fill of the io buffer between spans, timing read per launch)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, klang_amd

def run(K, N, B, amp, fill, reps=6):
    bank = klang_amd.FxBank("pingpong", K, max_block=N)
    torch.cuda.set_stream(torch.cuda.Stream())
    st = torch.cuda.current_stream().cuda_stream
    io = (torch.rand((B, K, 2, N), device="cuda") - 0.5) * amp
    for _ in range(max(2, 128 // B)): bank.render_device(io.data_ptr(), B, N, st)
    per = []
    for _ in range(reps):
        if fill: io.zero_()
        torch.cuda.synchronize(); bank.timing_begin()
        bank.render_device(io.data_ptr(), B, N, st)
        torch.cuda.synchronize()
        l, ms = bank.timing_end()
        per.append(1e3 * ms / B)
    bank.close()
    print(json.dumps(dict(K=K, blocks_per_span=B, input_amplitude=amp, fill_between_spans=fill, us_per_block=[round(x, 2) for x in per])), flush=True)

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for B in (64, 75, 128, 256):
    run(K, 256, B, 0.1, False)
run(K, 256, 75, 0.0, False)
run(K, 256, 75, 0.0, True)
run(K, 256, 75, 0.1, True)
