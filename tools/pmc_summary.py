#!/usr/bin/env python3
"""Aggregate rocprofv3 CSV output (kernel trace / counter collection) per kernel name -> small text/JSON.
usage: pmc_summary.py <rocprof-output-dir> [name-filter]"""
import csv, glob, json, os, sys
from collections import defaultdict

d = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
out = {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if filt and filt not in n:
            continue
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg[n[:90]]; a[0] += 1; a[1] += dur
    out["kernel_trace"] = {k: {"calls": v[0], "total_us": round(v[1], 1), "avg_us": round(v[1] / v[0], 2)} for k, v in agg.items()}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if filt and filt not in n:
            continue
        agg[n[:90]][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[n[:90]].add(r["Dispatch_Id"])
    out.setdefault("counters", {})
    for k, v in agg.items():
        nd = max(1, len(cnt[k]))
        out["counters"].setdefault(k, {"dispatches": nd})
        out["counters"][k].update({c: round(x / nd, 1) for c, x in v.items()})  # per-dispatch mean
print(json.dumps(out, indent=1))
