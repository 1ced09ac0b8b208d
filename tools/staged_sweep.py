#!/usr/bin/env python3
"""tools/staged_sweep.py pingpong|reverb K [G,C[,skipmask] ...] — kernel time of a recorded effect's staged form (klg_graph_staged.hpp) for workgroup shapes
G instances x C samples, optionally with levels left out (KLG_FX_STAGED_SKIP: wrong output, what the remaining levels cost).  One JSON line per shape."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, klang_amd

def main():
    which, K = sys.argv[1], int(sys.argv[2])
    name = {"pingpong": "pingpong_recorded", "reverb": "reverb_recorded"}[which]
    prog = open(os.path.join(ROOT, "tests", "golden", name + ".klgg")).read()
    rec = np.array([int(w, 16) for w in open(os.path.join(ROOT, "tests", "golden", name + ".rec")).read().split()], np.uint32)
    N = 256
    torch.cuda.set_stream(torch.cuda.Stream())
    st = torch.cuda.current_stream().cuda_stream
    for shape in sys.argv[3:] or ["16,32"]:
        p = shape.split(",")
        os.environ["KLG_FX_STAGED_G"], os.environ["KLG_FX_STAGED_C"] = p[0], p[1]
        os.environ["KLG_FX_STAGED_SKIP"] = p[2] if len(p) > 2 else "0"
        os.environ["KLG_FX_STAGED_LDS"] = str(160 * 1024)
        try:
            bank = klang_amd.FxBank(prog, K, max_block=N, initial_record=rec, channels=2)
        except Exception as e:
            print(json.dumps(dict(shape=shape, error=str(e)[:300]))); continue
        form = bank.graph_form()
        io = torch.rand((K, 2, N), device="cuda") - 0.5
        for _ in range(10): bank.process_device(io.data_ptr(), N, st)
        torch.cuda.synchronize(); bank.timing_begin()
        for _ in range(40): bank.process_device(io.data_ptr(), N, st)
        torch.cuda.synchronize()
        l, ms = bank.timing_end()
        print(json.dumps(dict(effect=which, K=K, shape=shape, form=form, kernel_ms=ms / l)), flush=True)
        bank.close()

if __name__ == "__main__":
    main()
