import sys, time, torch
sys.path.insert(0, "/root/repo")
import klang_amd
for patch in ("pingpong", "reverb"):
    K, N = 4096, 256
    bank = klang_amd.FxBank(patch, K, max_block=N)
    io = torch.zeros((K, 2, N), device="cuda")
    ts = torch.cuda.Stream()
    with torch.cuda.stream(ts):
        st = ts.cuda_stream
        for _ in range(20): bank.process_device(io.data_ptr(), N, st)
        torch.cuda.synchronize()
        for mode in ("plain", "timing", "zero+timing"):
            if mode != "plain": bank.timing_begin()
            t0 = time.perf_counter()
            for _ in range(300):
                if mode.startswith("zero"): io.zero_()
                bank.process_device(io.data_ptr(), N, st)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if mode != "plain": l, ms = bank.timing_end()
            print(patch, mode, "host us/call", 1e6 * (t1 - t0) / 300, "wall us/step", 1e6 * (t2 - t0) / 300, flush=True)
    bank.close()
