#!/usr/bin/env python3
"""tools/pmc_any.py <kernel substring> <set1> [<set2> ...] -- <command...>
One rocprofv3 pass per counter set (a set = comma-separated counter names; counters only with --kernel-trace), then the mean of every counter
over the later half of the named kernel's launches, as one JSON object.  Run on the GPU box (inside gpurun)."""
import csv, glob, json, os, shutil, subprocess, sys, tempfile
i = sys.argv.index("--")
name, sets, cmd = sys.argv[1], sys.argv[2:i], sys.argv[i + 1:]
out = {"kernel": name}
for s in sets:
    d = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
    r = subprocess.run(["rocprofv3", "--pmc", *s.split(","), "--kernel-trace", "--output-format", "csv", "-d", d, "--", *cmd],
                       cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
    vals = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if name in row["Kernel_Name"]:
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    if not vals:
        out["failed:" + s] = (r.stderr or r.stdout)[-300:]
    for k, v in vals.items():
        v = v[len(v) // 2:]
        out[k] = sum(v) / len(v); out["launches"] = len(v)
    shutil.rmtree(d, ignore_errors=True)
print(json.dumps(out))
