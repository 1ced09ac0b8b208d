#!/usr/bin/env python3
"""tools/fx_span_bench.py [K] — an effect bank's time per 256-sample block when the blocks are submitted as spans (klg_fx_render_device) of 1 / 4 / 16 / 64 blocks:
the kernel's own duration (events attached to the dispatch) and the stream's wall time around the spans, per block.  PingPong (hand-written; the recorded form),
Reverb.  Dials at rest (the blocks before the timed ones let PingPong's smoothers converge; FX_SPAN_FRESH=r: the first r spans of a new bank instead).  One JSON line per (effect, span length)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, klang_amd

def recorded(name):
    prog = open(os.path.join(ROOT, "tests", "golden", name + ".klgg")).read()
    rec = np.array([int(w, 16) for w in open(os.path.join(ROOT, "tests", "golden", name + ".rec")).read().split()], np.uint32)
    return prog, rec

def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    N = 256
    torch.cuda.set_stream(torch.cuda.Stream())
    st = torch.cuda.current_stream().cuda_stream
    for effect in (sys.argv[2:] or ["pingpong", "pingpong_recorded", "reverb"]):
        for B in [int(x) for x in os.environ.get("FX_SPAN_BLOCKS", "1,4,16,64").split(",")]:
            if effect == "pingpong_recorded":
                prog, rec = recorded("pingpong_recorded"); bank = klang_amd.FxBank(prog, K, max_block=N, initial_record=rec, channels=2)
            else:
                bank = klang_amd.FxBank(effect, K, max_block=N)
            for spec in filter(None, os.environ.get("FX_SPAN_DIALS", "").split(",")):             # e.g. FX_SPAN_DIALS=2=0.5,3=0.5: vibrato on every instance
                c, v = spec.split("=")
                for k in range(K): bank.set_control(k, int(c), float(v))
            io = (torch.rand((B, K, 2, N), device="cuda") - 0.5) * (0.0 if os.environ.get("FX_SPAN_SILENCE") else 0.1)
            fresh = int(os.environ.get("FX_SPAN_FRESH", "0"))                                  # time the first `fresh` spans of a new bank: the smoothers still converging
            if fresh: bank.render_device(io.data_ptr(), min(B, 8), N, st)                     # (as bench.py's cfg-4 legs: eight untimed blocks — a new object's smoothers start from 0, a delay too near to run ahead)
            else:
                for _ in range(max(2, 128 // B)): bank.render_device(io.data_ptr(), B, N, st)  # converge
            reps = fresh or max(4, 256 // B)
            torch.cuda.synchronize(); bank.timing_begin()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): bank.render_device(io.data_ptr(), B, N, st)
            e1.record(); torch.cuda.synchronize()
            launches, ms = bank.timing_end()
            per = 312 if effect == "reverb" else 32
            kern = ms / (reps * B)
            print(json.dumps(dict(effect=effect, K=K, N=N, blocks_per_span=B, launches_per_span=launches / reps, kernel_us_per_block=1e3 * kern, stream_us_per_block=1e3 * e0.elapsed_time(e1) / (reps * B),
                                  hbm_frac_on_kernel_time=K * N * per / (kern * 1e-3) / 8e12)), flush=True)
            bank.close()

if __name__ == "__main__":
    main()
