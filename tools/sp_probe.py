import json, os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, klang_amd
from test_gpu_graph import SUB2A_PROGRAM, sub2a_to_graph
def timed(bank, N, steps=200, warmup=60):
    mix = torch.zeros((2, N), dtype=torch.float32, device="cuda")
    torch.cuda.set_stream(torch.cuda.Stream()); st = torch.cuda.current_stream().cuda_stream
    for _ in range(warmup): bank.process_device(mix.data_ptr(), N, st)
    torch.cuda.synchronize(); bank.timing_begin()
    for _ in range(steps): bank.process_device(mix.data_ptr(), N, st)
    torch.cuda.synchronize(); n, ms = bank.timing_end()
    return round(1e3 * ms / n, 2)
V, P = 1024, 128
rng = np.random.default_rng(7); pitches = rng.integers(36, 97, size=V)
hand = klang_amd.SynthBank("sub2a", synths=V // P, notes=P, max_block=256)
hand.note_on_many(np.arange(V) // P, pitches, np.full(V, 0.8, np.float32))
hand.process(np.zeros((2, 256), np.float32))
words = np.stack([sub2a_to_graph(hand.voice_download(v)) for v in range(V)])
out = {"hand": {N: timed(hand, N) for N in (64, 128, 256)}}
for form in ("1", "4", "0"):
    os.environ["KLG_GRAPH_SP"] = form
    bank = klang_amd.SynthBank(SUB2A_PROGRAM, synths=V // P, notes=P, max_block=256)
    bank.voices_upload(np.arange(V, dtype=np.int32), words)
    out["rec_sp" + form] = {N: timed(bank, N) for N in (64, 128, 256)}
    bank.close()
print(json.dumps(out))
