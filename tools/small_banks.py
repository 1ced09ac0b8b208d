#!/usr/bin/env python3
"""tools/small_banks.py — hand-written note patches at plug-in sizes (32 .. 4,096 voices) playing SURVEY 8(d)'s script: kernel time and stream time per 256-sample
block.  What a bank costs when it cannot fill the chip is one wave's chain through the block plus whatever the launch form adds; this is where an anomaly in the latter shows
(the one-launch form's combine: DESIGN.md §3d).  One JSON line per (patch, voices)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
for patch in (sys.argv[1:] or ["sub2a", "supersaw", "fm3", "fm4"]):
    for V in (32, 128, 512, 1024, 2048, 4096):
        try:
            r = bench.run_literal_script(patch, V, 256, "%s_%d" % (patch, V))
            print(json.dumps({"patch": patch, "voices": V, "kernel_us": round(r["kernel_ms_mean"] * 1e3, 1), "step_us": round(r["ms_per_step"] * 1e3, 1)}), flush=True)
        except Exception as e:
            print(json.dumps({"patch": patch, "voices": V, "error": str(e)[:200]}), flush=True)
