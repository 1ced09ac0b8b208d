#!/usr/bin/env python3
"""tools/kernel_summary.py <rocprofv3 output dir> <kernel name substring> [settle warmup steps [stats.csv]]
Summarises a `rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline` run for profiles/: launches of the named
kernel, their average / median duration, and the average over the launches of bench.py's timed region (after the `settle`
blocks that take the voices to sustain and the `warmup` steps) — the number roofline.kernel_ms of the bench line must agree with."""
import csv
import glob
import json
import os
import sys


def main():
    d, name = sys.argv[1], sys.argv[2]
    settle, warmup, steps = (int(x) for x in sys.argv[3:6]) if len(sys.argv) >= 6 else (24, 20, 200)
    trace = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
    rows = []
    per_kernel = {}                                                                           # name -> [calls, total ns] over EVERY process of the run (bench.py's children write their own .db)
    for db in sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)):      # rocprofv3's default (rocpd sqlite) output
        import sqlite3
        c = sqlite3.connect(db)
        for name_, start, end in c.execute("select name, start, end from kernels"):
            k = per_kernel.setdefault(name_, [0, 0])
            k[0] += 1; k[1] += int(end) - int(start)
            if name in name_:
                rows.append((int(start), int(end) - int(start), name_))
    if len(sys.argv) >= 7 and per_kernel:                                                     # also write the per-kernel stats table
        total = sum(v[1] for v in per_kernel.values()) or 1
        with open(sys.argv[6], "w") as f:
            f.write("Name,Calls,TotalDurationUs,AverageUs,Percentage\n")
            for n_, (calls, ns) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1]):
                f.write('"%s",%d,%.3f,%.3f,%.4f\n' % (n_, calls, ns / 1e3, ns / 1e3 / calls, 100.0 * ns / total))
    for t in trace:
        for r in csv.DictReader(open(t)):
            if name in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    dur = [r[1] for r in rows]
    timed = dur[settle + warmup:settle + warmup + steps]
    out = {"kernel": rows[0][2] if rows else name, "launches": len(dur), "avg_ns_all_launches": sum(dur) / max(1, len(dur)),
           "median_ns": sorted(dur)[len(dur) // 2] if dur else None, "avg_ns_timed_region": sum(timed) / max(1, len(timed)),
           "timed_region": f"launches {settle + warmup}..{settle + warmup + steps - 1} (after {settle} settling blocks and {warmup} warmup steps)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
