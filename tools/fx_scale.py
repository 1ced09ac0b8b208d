"""tools/fx_scale.py [patch K K ...]... — effect kernel time vs instance count (HIP-event kernel time via klg_fx_timing_*)."""
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
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(4): bank.process_device(io.data_ptr(), N, st)
    torch.cuda.synchronize(); bank.timing_begin(); t0 = time.perf_counter()
    steps = 20
    for _ in range(steps): bank.process_device(io.data_ptr(), N, st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    l, ms = bank.timing_end()
    bps = {"pingpong": 32, "reverb": 312}[patch]
    print(json.dumps(dict(patch=patch, K=K, kernel_ms=ms / l, inst_samples_per_s=K * N * steps / dt, alg_GBs=K * N * bps / (ms / l * 1e-3) / 1e9, finite=bool(torch.isfinite(io).all().item()))), flush=True)
    bank.close(); del io; torch.cuda.empty_cache()
