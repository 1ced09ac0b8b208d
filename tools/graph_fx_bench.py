#!/usr/bin/env python3
"""tools/graph_fx_bench.py — a recorded graph effect (the Echo program of tests/test_gpu_graph.py with a Delay<192000>) at bank
scale on one MI355X: kernel time (HIP events) and algorithmic bytes (ring write 4 + new ring row 4 + io 8 B per instance*sample)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, klang_amd
from test_gpu_graph import ECHO_PROGRAM
prog = ECHO_PROGRAM.replace("node 0 delay 4800", "node 0 delay 192000")
for K in [int(x) for x in sys.argv[1:]] or [4096, 65536]:
    N = 256
    bank = klang_amd.FxBank(prog, K, max_block=N, channels=1)
    g = torch.Generator(device="cuda").manual_seed(1)
    io = torch.rand((K, 1, N), device="cuda", generator=g) - 0.5
    torch.cuda.set_stream(torch.cuda.Stream())      # a stream of our own: handle 0 (torch's default) means "the bank's own stream" to the library, which torch's clears are not ordered with
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(4): bank.process_device(io.data_ptr(), N, st)
    torch.cuda.synchronize(); bank.timing_begin(); t0 = time.perf_counter()
    for _ in range(20): bank.process_device(io.data_ptr(), N, st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    l, ms = bank.timing_end()
    print(json.dumps(dict(effect="graph echo (Delay<192000>)", K=K, kernel_ms=ms / l, inst_samples_per_s=K * N * 20 / dt, alg_GBs=K * N * 16 / (ms / l * 1e-3) / 1e9)), flush=True)
    bank.close(); del io; torch.cuda.empty_cache()
