#!/usr/bin/env python3
"""tools/pingpong_leg_repeat.py [runs] — bench.py's cfg-4 PingPong leg (4,096 instances, 375 blocks from a fresh bank: 8 untimed blocks, three 25-block spans, one 300-block
span) several times in ONE process on one box: how far one run's roofline fraction is from the next.  One line per run: kernel us per block over the whole script, its
fraction of HBM peak, and the same for the 300-block launch alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
os.environ["KLG_BENCH_PMC_FX"] = "0"
for run in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    r = bench.run_fx("pingpong", 4096, 256)
    ro = r["roofline"]
    print(run, round(r["kernel_ms_mean"] * 1e3, 3), round(ro["frac"], 4), round(ro["after_the_first_fifth"]["kernel_ms_mean"] * 1e3, 3), round(ro["after_the_first_fifth"]["frac"], 4), flush=True)
