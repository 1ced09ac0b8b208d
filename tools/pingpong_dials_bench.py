#!/usr/bin/env python3
"""tools/pingpong_dials_bench.py [K] — the hand-written PingPong kernel with one instance in seven off its default dials, one kind of deviation at a time:
near taps (Delay dial 0: the right tap 24 samples behind the cursor), vibrato (Scratch / Rate up), a moved Delay dial (the scratch detector fires while the
smoother is on its way), all of them at random.  Kernel time per 256-sample block right after the dials moved (blocks 10 - 50) and settled (blocks 300 - 340).
One JSON line per case."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, klang_amd

def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    N = 256
    torch.cuda.set_stream(torch.cuda.Stream())
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(3)
    cases = {
        "default": [],
        "near taps": [(k, c, v) for k in range(0, K, 7) for c, v in ((5, 0.0), (1, 0.001))],
        "vibrato": [(k, c, v) for k in range(0, K, 7) for c, v in ((2, 0.5), (3, 0.5))],
        "delay dial moved": [(k, c, v) for k in range(0, K, 7) for c, v in ((5, 0.31),)],
        "random": [(k, c, float(rng.uniform(lo, hi))) for k in range(0, K, 7) for c, lo, hi in ((0, 0.2, 0.9), (1, 0.01, 0.6), (2, 0.0, 1.0), (3, 0.01, 1.0), (5, 0.0, 0.4))],
    }
    for name, ctl in cases.items():
        bank = klang_amd.FxBank("pingpong", K, max_block=N)
        io = (torch.rand((K, 2, N), device="cuda") - 0.5) * 0.1
        for _ in range(60): bank.process_device(io.data_ptr(), N, st)          # the fresh bank's own smoothers settle
        for k, c, v in ctl: bank.set_control(k, c, v)
        out = {"case": name, "K": K}
        done = 0
        for label, first, count in (("moved_us_per_block", 10, 40), ("settled_us_per_block", 300, 40)):
            while done < first: bank.process_device(io.data_ptr(), N, st); done += 1
            torch.cuda.synchronize(); bank.timing_begin()
            for _ in range(count): bank.process_device(io.data_ptr(), N, st)
            done += count
            torch.cuda.synchronize()
            l, ms = bank.timing_end()
            out[label] = 1e3 * ms / l
        print(json.dumps(out), flush=True)
        bank.close()

if __name__ == "__main__":
    main()
