#!/usr/bin/env python3
"""tools/short_lines_bench.py — tests/patches/fx_short.k as the facade records it (two feedback lines read INSIDE a chunk of the sample-parallel kernel), a bank
of K instances at one Length: kernel time per 256-sample block of the staged form with the parts of a failed chunk (the product), with KLG_FX_STAGED_RETRY=0
(a failed chunk straight to the plain body: round 4) and of the one-lane-per-instance kernel; the three must agree bit for bit.  One JSON line per Length."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, klang_amd

PROGRAM = """klgg 1
kind effect 1
ctl 4
dial 0 0 0.899999976 0.600000024
dial 1 0.200000003 40 6
dial 2 0 0.899999976 0.300000012
dial 3 0.0500000007 1.60000002 0.5
node 0 delay 2400
node 1 delay 2400
node 2 fsine
node 3 lpf
node 4 smooth
op const 7 -1 -1 -1 42200000
op const 8 -1 -1 -1 3f3504f3
op lpfset -1 7 8 3 00000001
op ctl 11 -1 -1 -1 00000000
op ctl 12 -1 -1 -1 00000001
op ctl 13 -1 -1 -1 00000002
op in 16 -1 -1 -1 00000000
op const 20 -1 -1 -1 3f800000
op const 23 -1 -1 -1 473b8000
op const 25 -1 -1 -1 447a0000
op const 29 -1 -1 -1 44048000
op const 38 -1 -1 -1 3f000000
op smooth 17 -1 -1 4 00000003
op oscset -1 12 -1 2 00000000
op osc 18 -1 -1 2 00000000
op mul 19 18 13 -1 00000000
op add 21 20 19 -1 00000000
op mul 22 17 21 -1 00000000
op mul 24 22 23 -1 00000000
op div 26 24 25 -1 00000000
op delaytap 27 26 -1 0 00000000
op mul 28 22 23 -1 00000000
op div 30 28 29 -1 00000000
op delaytap 31 30 -1 1 00000000
op mul 32 27 11 -1 00000000
op add 33 16 32 -1 00000000
op delayin -1 33 -1 0 00000000
op mul 34 31 11 -1 00000000
op sub 35 27 34 -1 00000000
op delayin -1 35 -1 1 00000000
op add 36 16 27 -1 00000000
op add 37 36 31 -1 00000000
op mul 39 37 38 -1 00000000
op lpf 40 39 -1 3 00000000
prepare 3
ret 40
end
"""
REC = np.array([int(w, 16) for w in "00000000 00000000 00000000 00000000 00000000 00000000 00000000 447a0000 3f800000 00000000 00000000 00000000 00000000 00000000 00000000 00000000 00000000 00000000".split()], np.uint32)


def main():
    K, N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 256
    torch.cuda.set_stream(torch.cuda.Stream())
    st = torch.cuda.current_stream().cuda_stream
    for length_ms in (0.1, 0.25, 0.45, 0.8, 1.6):                       # x 48 samples: 4.8, 12, 21.6, 38, 77 samples behind the cursor (the second line 1.9 x that)
        res = {"effect": "tests/patches/fx_short.k (recorded)", "K": K, "N": N, "length_ms": length_ms, "first_tap_samples": length_ms * 48.0}
        outs = {}
        for name, env in (("staged_parts", {"KLG_FX_STAGED": "1", "KLG_FX_STAGED_RETRY": "1"}), ("staged_no_retry", {"KLG_FX_STAGED": "1", "KLG_FX_STAGED_RETRY": "0"}), ("one_lane", {"KLG_FX_STAGED": "0", "KLG_FX_STAGED_RETRY": "1"})):
            os.environ.update(env)
            bank = klang_amd.FxBank(PROGRAM, K, max_block=N, initial_record=REC, channels=1)
            for k in range(K):
                bank.set_control(k, 3, length_ms); bank.set_control(k, 2, 0.1)
            g = torch.Generator(device="cuda").manual_seed(1)
            io = torch.rand((K, 1, N), device="cuda", generator=g) - 0.5
            for _ in range(40): bank.process_device(io.data_ptr(), N, st)     # the smoothed Length arrives (1,000-sample time constant)
            torch.cuda.synchronize(); bank.timing_begin()
            for _ in range(20): bank.process_device(io.data_ptr(), N, st)
            torch.cuda.synchronize()
            l, ms = bank.timing_end()
            res[name + "_ms"] = ms / l
            outs[name] = io.clone()
            bank.close()
        res["bit_identical"] = bool((outs["staged_parts"].view(torch.int32) == outs["one_lane"].view(torch.int32)).all().item() and (outs["staged_no_retry"].view(torch.int32) == outs["one_lane"].view(torch.int32)).all().item())
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
