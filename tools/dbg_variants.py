import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import klang_amd
from test_gpu_graph import SUB2A_PROGRAM, sub2a_to_graph
V, P, N = int(os.environ.get('VOICES', 1 << 18)), 128, 256
rng = np.random.default_rng(20250314)
pitches = rng.integers(36, 97, size=V)
proto = klang_amd.SynthBank("sub2a", synths=1, notes=61, max_block=N)
for p in range(36, 97): proto.note_on(0, p, 0.8)
recs = {36 + i: sub2a_to_graph(proto.voice_download(i)) for i in range(61)}
proto.close()
outs = {}
for variant in ("hand_x2", "hand_x1", "graph_x2", "graph_x1"):
    os.environ.pop("KLG_RENDER_X1", None); os.environ.pop("KLG_GRAPH_X1", None)
    if variant == "hand_x1": os.environ["KLG_RENDER_X1"] = "1"
    if variant == "graph_x1": os.environ["KLG_GRAPH_X1"] = "1"
    if variant.startswith("graph"):
        bank = klang_amd.SynthBank(SUB2A_PROGRAM, synths=V // P, notes=P, max_block=N)
        words = np.stack([recs[int(p)] for p in pitches])
        for c0 in range(0, V, 1 << 16):
            bank.voices_upload(np.arange(c0, min(V, c0 + (1 << 16)), dtype=np.int32), words[c0:c0 + (1 << 16)])
    else:
        bank = klang_amd.SynthBank("sub2a", synths=V // P, notes=P, max_block=N)
        bank.note_on_many(np.arange(V) // P, pitches, np.full(V, 0.8, np.float32))
    for b in range(int(os.environ.get("BLOCKS", "35"))):
        mix = np.zeros((2, N), np.float32); bank.process(mix)
    outs[variant] = mix.copy(); bank.close()
ref = outs["hand_x2"]
for k, v in outs.items():
    print(k, float(np.abs(v).sum()), float(np.abs(v - ref).max() / np.abs(ref).max()))
