#!/usr/bin/env python3
"""tools/staged_examples_bench.py — every example effect of the facade fixtures as a bank of K instances: kernel time per 256-sample block of the staged
(sample-parallel) generated kernel and of the one-lane-per-instance one, and the algorithmic HBM bytes the program implies (io 8 B per channel and
sample + per delay node one row written per input() and ~one new row read per tap).  The programs are what the facade records from the unchanged .k files
(KLANG_MI355_DUMP_GRAPH=1 while it runs the fixture)."""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch, klang_amd
from test_gpu_fx_facade import NAMES, run_effect
import pathlib


def program_of(name):
    os.environ["KLANG_MI355_FORCE_GRAPH"] = "1"; os.environ["KLANG_MI355_DUMP_GRAPH"] = "1"
    with tempfile.TemporaryDirectory() as d:
        log = os.path.join(d, "err.txt")
        saved = os.dup(2); f = os.open(log, os.O_WRONLY | os.O_CREAT); os.dup2(f, 2)
        try:
            run_effect(name, pathlib.Path(d))
        finally:
            os.dup2(saved, 2); os.close(f); os.close(saved)
        err = open(log).read()
    m = re.search(r"^klgg 1\n.*?^end\n", err, re.S | re.M)
    rec = re.search(r"initial record:((?: [0-9a-f]{8})+)", err)
    return (m.group(0) if m else None), (np.array([int(w, 16) for w in rec.group(1).split()], np.uint32) if rec else None)


def timed(bank, io, N, st, steps=30):
    for _ in range(6): bank.process_device(io.data_ptr(), N, st)
    torch.cuda.synchronize(); bank.timing_begin()
    for _ in range(steps): bank.process_device(io.data_ptr(), N, st)
    torch.cuda.synchronize()
    l, ms = bank.timing_end()
    return ms / l


def main():
    N = 256
    Ks = [int(x) for x in sys.argv[1:]] or [4096, 65536]
    torch.cuda.set_stream(torch.cuda.Stream()); st = torch.cuda.current_stream().cuda_stream
    for name in NAMES + ["fx_topchorus"]:
        prog, rec = program_of(name)
        if prog is None:
            print(json.dumps(dict(effect=name, error="no program")), flush=True); continue
        ch = 2 if "kind effect 2" in prog else 1
        ops = [l.split() for l in prog.splitlines() if l.startswith("op ")]
        nprep = int(re.search(r"^prepare (\d+)", prog, re.M).group(1)) if re.search(r"^prepare (\d+)", prog, re.M) else 0
        body = ops[nprep:]
        writes = sum(1 for o in body if o[1] == "delayin"); reads = sum(1 for o in body if o[1] in ("delaytap", "delayout"))
        bytes_per = 8 * ch + 4 * writes + 4 * reads
        for K in Ks:
            ring = sum(int(l.split()[3]) + 1 for l in prog.splitlines() if l.startswith("node ") and l.split()[2] == "delay") * 4 * K
            if ring > 150e9: continue
            row = dict(effect=name, K=K, channels=ch, ops_per_sample=len(body), delay_writes=writes, delay_reads=reads, alg_bytes_per_instance_sample=bytes_per)
            for form, env in (("staged", "1"), ("one_lane", "0")):
                if form == "one_lane" and K > 16384: continue
                os.environ["KLG_FX_STAGED"] = env
                bank = klang_amd.FxBank(prog, K, max_block=N, initial_record=rec, channels=ch)
                if form == "staged": row["form"] = bank.graph_form()
                io = (torch.rand((K, ch, N), device="cuda") - 0.5) * 0.1
                ms = timed(bank, io, N, st)
                row[form + "_ms"] = ms
                if form == "staged": row["staged_alg_TBps"] = K * N * bytes_per / (ms * 1e-3) / 1e12
                bank.close(); del io, bank; torch.cuda.empty_cache()
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
