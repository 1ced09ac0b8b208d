"""tools/fx_scale.py [patch K K ...]... — effect kernel time vs instance count (HIP-event kernel time via klg_fx_timing_*).  PingPong is warmed for 300 blocks
(KLG_FX_WARM): the steady state of a plugin whose dials are not being turned (its control smoothers take some ten thousand samples to settle)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, klang_amd
jobs, patch = [], None
for a in sys.argv[1:]:
    if a.isdigit(): jobs.append((patch, int(a)))
    else: patch = a
for patch, K in jobs:
    N = int(os.environ.get("KLG_FX_N", "256"))
    bank = klang_amd.FxBank(patch, K, max_block=N)
    g = torch.Generator(device="cuda").manual_seed(1)
    io = (torch.rand((K, 2, N), device="cuda", generator=g) - 0.5)
    ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream       # (not the null stream: its implicit synchronisation shows up in the launch gaps)
    for _ in range(int(os.environ.get("KLG_FX_WARM", "300" if patch == "pingpong" else "4"))): bank.process_device(io.data_ptr(), N, st)
    torch.cuda.synchronize(); bank.timing_begin(); t0 = time.perf_counter()
    steps = 20
    for _ in range(steps): bank.process_device(io.data_ptr(), N, st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    l, ms = bank.timing_end()
    bps = {"pingpong": 32, "reverb": 312}[patch]
    print(json.dumps(dict(patch=patch, K=K, kernel_ms=ms / l, inst_samples_per_s=K * N * steps / dt, alg_GBs=K * N * bps / (ms / l * 1e-3) / 1e9, finite=bool(torch.isfinite(io).all().item()))), flush=True)
    bank.close(); del io; torch.cuda.empty_cache()
