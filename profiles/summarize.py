#!/usr/bin/env python3
"""Condense rocprofv3 output of profiles/collect.sh into small text/JSON summaries.

  <dir>/kt/**/*kernel_stats.csv          -> summary/kernel_stats.txt   (per-kernel calls / total / average)
  <dir>/pmc*/**/*counter_collection.csv  -> summary/pmc.json           (per kernel: mean counter value per dispatch)
  + summary/traffic.json: HBM bytes per launch of the loss kernel, corrected as MI355X_MICROARCH.md prescribes
    (FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B read requests as 64 B, i.e. it reports
    half the bytes of a wide stream -> doubled; the pack kernels, whose byte counts are known, give the measured
    calibration factor that is reported next to it).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("void ", "")[:70]


def main(d):
    out = os.path.join(d, "summary")
    os.makedirs(out, exist_ok=True)
    stats = glob.glob(os.path.join(d, "kt", "**", "*kernel_stats.csv"), recursive=True)
    lines = []
    for f in stats:
        rows = list(csv.DictReader(open(f)))
        lines.append("# %s" % os.path.relpath(f, d))
        lines.append("%-72s %8s %14s %12s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
        for r in rows:
            lines.append("%-72s %8s %14s %12.0f %8s" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"],
                                                       float(r["AverageNs"]), r["Percentage"]))
    open(os.path.join(out, "kernel_stats.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:14]))

    pmc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    summ = {k: {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()} for k, cs in pmc.items()}
    json.dump(summ, open(os.path.join(out, "pmc.json"), "w"), indent=1, sort_keys=True)
    loss = [k for k in summ if k.startswith("pcl_loss_kernel")]
    for k in loss:
        c = summ[k]
        print(k, {n: round(v["mean"], 1) for n, v in c.items()})
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            fetch, write = c["FETCH_SIZE"]["mean"] * 1024, c["WRITE_SIZE"]["mean"] * 1024
            tr = {"kernel": k, "fetch_size_bytes_raw": fetch, "write_size_bytes": write,
                  "hbm_bytes_per_launch": 2 * fetch + write,
                  "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide reads); "
                          "includes Infinity-Cache hits (memory-side request counters)"}
            json.dump(tr, open(os.path.join(out, "traffic_%s.json" % k.replace("<", "_").replace(">", "").replace(", ", "_")), "w"), indent=1)
            print("traffic", tr)
            # bench.py reads profiles/traffic.json: {workload: {"hbm_bytes_per_launch": ...}}
            wl = os.environ.get("WORKLOAD", "cfg2")
            merged = {}
            tj = os.path.join(out, "traffic.json")
            if os.path.exists(tj):
                merged = json.load(open(tj))
            merged[wl] = tr
            json.dump(merged, open(tj, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
