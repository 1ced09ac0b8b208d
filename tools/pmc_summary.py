#!/usr/bin/env python3
"""tools/pmc_summary.py <pmc dir (tools/pmc_run.sh output)> <kernel name substring> — mean FETCH_SIZE / WRITE_SIZE (KB) over the
launches of the named kernel and the HBM bytes per launch, corrected as MI355X_MICROARCH.md prescribes for gfx950
(FETCH_SIZE counts half: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024).  Prints a JSON object for profiles/pmc_traffic.json."""
import csv
import glob
import json
import os
import sys


def main():
    d, name = sys.argv[1], sys.argv[2]
    tot = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if name in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                tot.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    out = {"kernel": name}
    for k, v in tot.items():
        v = v[len(v) // 4:]                                        # skip the settling launches (ramps: same traffic, but be safe)
        out[k + "_KB"] = sum(v) / len(v); out["launches"] = len(v)
    if "FETCH_SIZE_KB" in out and "WRITE_SIZE_KB" in out:
        out["bytes_per_launch"] = (2 * out["FETCH_SIZE_KB"] + out["WRITE_SIZE_KB"]) * 1024
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
