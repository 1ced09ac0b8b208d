"""Debug helper (GPU box): small patches recorded by the facade vs the genuine reference's output computed in the build container."""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
for n in sys.argv[1:]:
    scn = os.path.join(ROOT, "tools", f"_dbg_{n}.scn"); ref = np.load(os.path.join(ROOT, "tools", f"_dbg_{n}.npz"))
    out = f"/tmp/_dbg_{n}.bin"
    subprocess.run([os.path.join(ROOT, "tools", f"_dbg_facade_{n}"), scn, out], check=True)
    d = open(out, "rb").read()
    magic, N, B, P = (int(x) for x in np.frombuffer(d, np.int32, 4))
    mix = np.frombuffer(d, np.float32, B * 2 * N, 16).reshape(B, 2, N)
    r = ref["mix"]
    bad = np.argwhere(mix.view(np.uint32) != r.view(np.uint32))
    print(n, "max abs err", float(np.abs(mix - r).max()), "first bad", bad[0] if len(bad) else None, "n bad", len(bad), "of", mix.size)
    if len(bad):
        b, c, i = bad[0]
        print("   got", mix[b, c, i:i + 6], "\n   ref", r[b, c, i:i + 6])
