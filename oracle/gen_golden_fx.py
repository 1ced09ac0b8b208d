"""oracle/gen_golden_fx.py — TEST INFRASTRUCTURE: effect scenarios (BASELINE config 4) for gen_golden.py."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from scenario_io import Scenario  # noqa: E402


def fx_scenarios():
    out = {}
    # PingPong.k, default controls: noise burst then silence, 2 instances with different inputs
    s = Scenario(patch="pingpong", block=256, blocks=60, instances=2, burst=4800, seed=11, dump=[0, 1, 5, 18, 19, 40, 59])
    out["pingpong_default"] = s
    # PingPong.k, "Funky Beat" preset (PingPong.k:25) + a delay-dial move mid-run (exercises the scratch branch + LFO)
    s = Scenario(patch="pingpong", block=256, blocks=60, instances=2, burst=9600, seed=12, dump=[0, 1, 10, 30, 31, 45, 59])
    s.ctl = [(0, 0.663), (1, 0.248), (2, 0.411), (3, 0.594), (4, 2.000), (5, 0.283)]
    s.control(30, 0, 5, 0.05)
    s.control(30, 1, 5, 0.6)
    out["pingpong_preset"] = s
    # block-size independence: N = 64
    s = Scenario(patch="pingpong", block=64, blocks=80, instances=1, burst=2000, seed=13, dump=[0, 40, 79])
    s.ctl = [(0, 0.9), (1, 0.02), (5, 0.02)]
    out["pingpong_n64"] = s
    # Reverb.k defaults (Early only) and the "Large Hall" preset (Reverb.k:116): all four LateReflections audible
    s = Scenario(patch="reverb", block=256, blocks=40, instances=2, burst=4800, seed=21, dump=[0, 1, 10, 20, 39])
    out["reverb_default"] = s
    s = Scenario(patch="reverb", block=256, blocks=40, instances=2, burst=4800, seed=22, dump=[0, 1, 10, 20, 39])
    s.ctl = [(0, 1.0), (1, 0.0), (2, 0.419), (3, 0.329), (4, 1.0), (5, 10.0), (6, 100.0), (7, 0.5), (8, 0.5), (9, 0.1)]
    out["reverb_large_hall"] = s
    # control change mid-run -> prepare() re-runs reflections.set() with the delay lines in flight
    s = Scenario(patch="reverb", block=256, blocks=30, instances=1, burst=3000, seed=23, dump=[0, 14, 15, 16, 29])
    s.ctl = [(1, 0.7), (2, 0.5), (3, 0.5), (6, 0.4)]
    s.control(15, 0, 6, 0.9)
    s.control(15, 0, 7, 0.3)
    out["reverb_retune"] = s
    return out
