"""Debug helper: which PingPong control regime differs from the oracle (run on the GPU box)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from scenario_io import Scenario
from klg_driver import run_scenario_oracle, run_fx_scenario_gpu, rel_err, bit_exact_fraction
import subprocess
root = os.path.join(os.path.dirname(__file__), "..")
subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "oracle"])
OBDIR = os.path.join(root, "oracle", "_build")
def variant(name, setup, K=8, blocks=12):
    s = Scenario(patch="pingpong", block=192, blocks=blocks, instances=K, burst=2500, seed=11, dump=list(range(blocks)))
    setup(s, K)
    ref = run_scenario_oracle(s, OBDIR)["per_voice"]
    got = run_fx_scenario_gpu(s)["per_voice"]
    bad = np.argwhere(ref.view(np.uint32) != got.view(np.uint32))
    print(name, "rel", rel_err(got, ref), "exact", bit_exact_fraction(got, ref), "first bad", bad[0] if len(bad) else None, flush=True)
    if len(bad):
        b = tuple(bad[0]); print("   ref", ref[b], "got", got[b])
def short(s, K):
    for k in range(K): s.control(0, k, 1, [0.001, 0.0012, 0.0016, 0.002, 0.004, 0.3, 0.01, 0.05][k % 8])
def vib(s, K):
    for k in range(K): s.control(0, k, 2, 0.1 + 0.1 * k); s.control(0, k, 3, 0.05 + 0.1 * k)
def vib_short(s, K): short(s, K); vib(s, K)
def glide(s, K):
    short(s, K)
    for k in range(0, K, 3): s.control(4, k, 1, 0.5); s.control(8, k, 1, 0.001)
def c4(s, K):
    for k in range(K): s.control(0, k, 4, 0.1 * k)
def c0(s, K):
    for k in range(K): s.control(0, k, 0, 0.3 + 0.08 * k)
for nm, f in [("short", short), ("vib", vib), ("vib_short", vib_short), ("glide", glide), ("c4", c4), ("c0", c0)]:
    variant(nm, f)
def full(s, K):
    rng = np.random.default_rng(9)
    for k in range(K):
        s.control(0, k, 0, float(rng.uniform(0.3, 0.95)))
        s.control(0, k, 1, float(rng.choice([0.001, 0.0012, 0.0016, 0.002, 0.004, 0.3])))
        s.control(0, k, 2, float(rng.uniform(0.0, 1.0)))
        s.control(0, k, 3, float(rng.uniform(0.01, 1.0)))
        s.control(0, k, 4, float(rng.uniform(0.0, 1.0)))
s = Scenario(patch="pingpong", block=192, blocks=20, instances=130, burst=2500, seed=11, dump=list(range(20)))
full(s, 130)
ref = run_scenario_oracle(s, OBDIR)["per_voice"]; got = run_fx_scenario_gpu(s)["per_voice"]
print(ref.shape)
neq = (ref.view(np.uint32) != got.view(np.uint32))
per_inst = neq.reshape(neq.shape[0], 130, -1).any(axis=2)      # [B][K]
badk = np.argwhere(per_inst.any(axis=0)).ravel()
print("bad instances", badk)
ctl = {}
for e in s.ev:
    if e[1] == 2 or True:
        ctl.setdefault(e[2], {})[int(e[3])] = e[4]
for k in badk[:10]:
    fb = np.argwhere(per_inst[:, k]).ravel()[0]
    idx = np.argwhere(neq[fb, k])[0]
    print(k, ctl.get(int(k)), "first bad block", fb, "ch,sample", idx, ref[fb, k][tuple(idx)], got[fb, k][tuple(idx)])
