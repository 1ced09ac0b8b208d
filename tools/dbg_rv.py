import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from scenario_io import Scenario
from klg_driver import run_fx_scenario_gpu, run_scenario_oracle
base = [(0, 1.0), (1, 0.0), (2, 0.419), (3, 0.329), (4, 1.0), (5, 10.0), (6, 100.0), (7, 0.5), (8, 0.5), (9, 0.1)]
for drop in (None, 0, 6, 7, 8, 3):
    s = Scenario(patch="reverb", block=256, blocks=24, instances=1, burst=3000, seed=22, dump=list(range(24)))
    s.ctl = [c for c in base if c[0] != drop]
    ref = run_scenario_oracle(s, '/root/repo/oracle/_build')["per_voice"]
    got = run_fx_scenario_gpu(s)["per_voice"]
    bad = np.argwhere(got.view(np.uint32)!=ref.view(np.uint32))
    print('drop', drop, 'max ref', np.abs(ref).max(), 'max diff', np.abs(got-ref).max(), 'nbad', len(bad), bad[:2].tolist())
