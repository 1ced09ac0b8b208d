#!/usr/bin/env python3
"""tools/fm_leg.py [voices ...] — bench.py's cfg-5 leg (FM.k with a fourth operator; the script's 375 blocks) at the given bank sizes (default 131072, 1048576), one line each:
kernel ms per block, fraction of the fp32 peak on the sounding voices, and the wall time per block."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for v in [int(x) for x in sys.argv[1:]] or [131072, 1048576]:
    r = bench.run_literal_script("fm4", v, 256, f"fm4_{v}")
    print(json.dumps({"voices": v, "kernel_ms_mean": r.get("kernel_ms_mean"), "frac": r["roofline"]["frac"], "ms_per_step": r.get("ms_per_step"), "value": r.get("value")}), flush=True)
