#!/usr/bin/env python3
"""tools/sub2a_bench.py [voices] — config 2's patch playing SURVEY 8(d)'s script (bench.run_literal_script): value, ms per block, kernel ms.
KLG_SUB2A_SP = 1 / 0 forces the one-voice-per-wave kernel (klg_render_sub2a_sp, default up to 2,048 voices) / the packed two-voices-per-lane kernel."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
r = bench.run_literal_script("sub2a", V, 256, "sub2a_%d_sp%s" % (V, os.environ.get("KLG_SUB2A_SP", "default")))
print(json.dumps({k: r.get(k) for k in ("name", "value", "ms_per_step", "kernel_ms_mean", "ms_per_block_sustain", "value_sustain_phase")}))
