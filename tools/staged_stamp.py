import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["KLG_FX_STAGED_STAMP"] = "1"
import torch, klang_amd
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
which, K = sys.argv[1], int(sys.argv[2])
name = {"pingpong": "pingpong_recorded", "reverb": "reverb_recorded"}[which]
prog = open(os.path.join(ROOT, "tests", "golden", name + ".klgg")).read()
rec = np.array([int(w, 16) for w in open(os.path.join(ROOT, "tests", "golden", name + ".rec")).read().split()], np.uint32)
bank = klang_amd.FxBank(prog, K, max_block=256, initial_record=rec, channels=2)
io = torch.rand((K, 2, 256), device="cuda") - 0.5
for _ in range(4):
    bank.process_device(io.data_ptr(), 256, None)
    bank.sync()
