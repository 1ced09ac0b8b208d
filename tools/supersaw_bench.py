#!/usr/bin/env python3
"""tools/supersaw_bench.py [voices] — SuperSaw.k (config 3) playing SURVEY 8(d)'s script (bench.run_literal_script): value, ms per block, kernel ms.
KLG_SUPERSAW_LANES = 3 (default: klg_render_supersaw_sp), 2 (klg_render_supersaw_pairs), 1, 0 (a voice per lane) selects the kernel."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
r = bench.run_literal_script("supersaw", V, 256, "supersaw_%d_lanes%s" % (V, os.environ.get("KLG_SUPERSAW_LANES", "default")))
print(json.dumps({k: r[k] for k in ("name", "value", "ms_per_step", "kernel_ms_mean", "ms_per_block_sustain", "value_sustain_phase")}))
