#!/usr/bin/env python3
"""tools/reverb_recorded_bench.py [K ...] — the shipped examples/Reverb.k as a RECORDED graph effect (tests/golden/reverb_recorded.klgg + .rec: the program the
facade records from the unchanged file and the record of an instance after its host-run prepare(), dumped with KLANG_MI355_FORCE_GRAPH=1
KLANG_MI355_DUMP_GRAPH=1) against the hand-written kernel: kernel time of each per 256-sample block (HIP events).  Parity of the recorded form is
tests/test_gpu_fx_facade.py::test_shipped_reverb_k_recorded_as_a_graph; this is its price.  One JSON line per bank size."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, klang_amd
PROGRAM = open(os.path.join(ROOT, "tests", "golden", "reverb_recorded.klgg")).read()
RECORD = np.array([int(w, 16) for w in open(os.path.join(ROOT, "tests", "golden", "reverb_recorded.rec")).read().split()], np.uint32)


def timed(bank, io, N, steps=20, warmup=4):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(warmup): bank.process_device(io.data_ptr(), N, st)
    torch.cuda.synchronize(); bank.timing_begin()
    for _ in range(steps): bank.process_device(io.data_ptr(), N, st)
    torch.cuda.synchronize()
    l, ms = bank.timing_end()
    return ms / l


def main():
    N = 256
    torch.cuda.set_stream(torch.cuda.Stream())                   # one stream for torch's work and the library's launches (handle 0 would mean "the bank's own")
    for K in [int(x) for x in sys.argv[1:]] or [256, 1024]:
        io = torch.rand((K, 2, N), device="cuda") - 0.5
        out = {"K": K, "record_words": int(RECORD.size)}
        def recorded(staged):
            os.environ["KLG_FX_STAGED"] = "1" if staged else "0"          # (read when the program is compiled; the two forms are cached separately)
            return klang_amd.FxBank(PROGRAM, K, max_block=N, initial_record=RECORD, channels=2)
        res = {}
        for name, make in (("hand_written_ms", lambda: klang_amd.FxBank("reverb", K, max_block=N)),
                           ("recorded_graph_ms", lambda: recorded(True)), ("recorded_one_lane_per_instance_ms", lambda: recorded(False))):
            if name == "recorded_one_lane_per_instance_ms" and os.environ.get("KLG_BENCH_SKIP_PLAIN"): continue
            bank = make()
            if name == "recorded_graph_ms": out["form"] = bank.graph_form()
            x = io.clone()
            for _ in range(6): bank.process_device(x.data_ptr(), N, torch.cuda.current_stream().cuda_stream)   # six blocks of the bank's own output fed back: what the forms are compared on
            torch.cuda.synchronize(); res[name] = x.clone()
            out[name] = timed(bank, io.clone(), N)
            bank.close(); torch.cuda.empty_cache()
        if "recorded_one_lane_per_instance_ms" in res: out["forms_bit_identical"] = bool(torch.equal(res["recorded_graph_ms"].view(torch.int32), res["recorded_one_lane_per_instance_ms"].view(torch.int32)))
        out["recorded_equals_hand_written"] = bool(torch.equal(res["recorded_graph_ms"].view(torch.int32), res["hand_written_ms"].view(torch.int32)))
        out["ratio"] = out["recorded_graph_ms"] / out["hand_written_ms"]
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
