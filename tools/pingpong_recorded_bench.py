#!/usr/bin/env python3
"""tools/pingpong_recorded_bench.py — the shipped examples/PingPong.k as a RECORDED graph effect (tests/golden/pingpong_recorded.klgg + .rec: what the
facade records from the unchanged file and the record a fresh object packs to, dumped with KLANG_MI355_FORCE_GRAPH=1 KLANG_MI355_DUMP_GRAPH=1) against its hand-written kernel
(klg_fx_pingpong_x): the same input through both banks must agree bit for bit; kernel time of each (HIP events).  One JSON line per bank size."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, klang_amd

PROGRAM = open(os.path.join(ROOT, "tests", "golden", "pingpong_recorded.klgg")).read()


def initial_record():
    """What the facade packs from a fresh PingPong object after prepare() (dumped beside the program): zeros but the LFO's defaults, the
    filters' b0 = 1 and the instance's own copy of controls[1] (0.5)."""
    return np.array([int(w, 16) for w in open(os.path.join(ROOT, "tests", "golden", "pingpong_recorded.rec")).read().split()], np.uint32)


def timed(bank, io, N, steps=40, warmup=int(os.environ.get("KLG_BENCH_WARM", "10"))):
    st = torch.cuda.current_stream().cuda_stream                  # (main() makes a non-default stream current: handle 0 would mean "the bank's own stream")
    for _ in range(warmup): bank.process_device(io.data_ptr(), N, st)
    torch.cuda.synchronize(); bank.timing_begin()
    for _ in range(steps): bank.process_device(io.data_ptr(), N, st)
    torch.cuda.synchronize()
    l, ms = bank.timing_end()
    return ms / l


def main():
    N = 256
    torch.cuda.set_stream(torch.cuda.Stream())                   # one stream for torch's work and the library's launches
    for K, dials in [(int(x), d) for x in (sys.argv[1:] or [4096, 16384]) for d in ("default", "random")]:
        rng = np.random.default_rng(3)
        def recorded(staged):
            os.environ["KLG_FX_STAGED"] = "1" if staged else "0"          # (read when the program is compiled; the two forms are cached separately)
            return klang_amd.FxBank(PROGRAM, K, max_block=N, initial_record=initial_record(), channels=2)
        # one bank at a time (65,536 instances are 100 GB of rings each)
        banks = {"hand-written klg_fx_pingpong_x": lambda: klang_amd.FxBank("pingpong", K, max_block=N),
                 "recorded graph (klg_fx_graph)": lambda: recorded(True), "recorded graph, one lane per instance": lambda: recorded(False)}
        form = None
        ctl = [] if dials == "default" else [(k, c, float(rng.uniform(lo, hi))) for k in range(0, K, 7) for c, lo, hi in ((0, 0.2, 0.9), (1, 0.01, 0.6), (2, 0.0, 1.0), (3, 0.01, 1.0), (5, 0.0, 0.4))]
        outs = {}
        for name, make in banks.items():
            bank = make()
            if name == "recorded graph (klg_fx_graph)": form = bank.graph_form()
            for k, c, v in ctl: bank.set_control(k, c, v)
            g = torch.Generator(device="cuda").manual_seed(1)
            res = []
            for b in range(12):
                io = (torch.rand((K, 2, N), device="cuda", generator=g) - 0.5) * (1.0 if b < 6 else 0.0)
                bank.process_device(io.data_ptr(), N, torch.cuda.current_stream().cuda_stream)
                res.append(io.clone())
            torch.cuda.synchronize()
            outs[name] = torch.stack(res)
            io = torch.zeros((K, 2, N), device="cuda")
            outs[name + " ms"] = timed(bank, io, N)
            bank.close(); del bank
            torch.cuda.empty_cache()
        a, b = outs["hand-written klg_fx_pingpong_x"], outs["recorded graph (klg_fx_graph)"]
        same = bool(torch.equal(a.view(torch.int32), b.view(torch.int32)))
        if not same and os.environ.get("KLG_BENCH_DEBUG"):
            d = (a.view(torch.int32) != b.view(torch.int32)).nonzero()
            print("differing", len(d), "of", a.numel(), "first", d[:5].tolist(), "max abs", float((a - b).abs().max()), "values", [(float(a[tuple(i)]), float(b[tuple(i)])) for i in d[:5]], file=sys.stderr)
        c = outs["recorded graph, one lane per instance"]
        same_lane = bool(torch.equal(a.view(torch.int32), c.view(torch.int32)))
        hand, rec = outs["hand-written klg_fx_pingpong_x ms"], outs["recorded graph (klg_fx_graph) ms"]
        print(json.dumps(dict(effect="examples/PingPong.k", K=K, N=N, dials=dials, bit_identical=same, one_lane_form_bit_identical=same_lane, form=form, peak=float(a.abs().max()), hand_written_kernel_ms=hand, recorded_kernel_ms=rec,
                              recorded_one_lane_per_instance_ms=outs["recorded graph, one lane per instance ms"],
                              recorded_vs_hand=rec / hand, hand_alg_TBps=K * N * 32 / (hand * 1e-3) / 1e12, recorded_alg_TBps=K * N * 32 / (rec * 1e-3) / 1e12)), flush=True)


if __name__ == "__main__":
    main()
