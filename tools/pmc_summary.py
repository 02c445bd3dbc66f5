#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel-name (and dispatch order) counter sums."""
import csv, sys, collections, glob
for d in sys.argv[1:]:
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        rows = list(csv.DictReader(open(f)))
        agg = collections.OrderedDict()
        for r in rows:
            key = (int(r["Dispatch_Id"]), r["Kernel_Name"][:60])
            agg.setdefault(key, {})[r["Counter_Name"]] = agg.setdefault(key, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        print("==", f)
        for (did, name), c in agg.items():
            if "pack" in name or "fill" in name.lower():
                continue
            print(did, name, " ".join(f"{k}={v:.4g}" for k, v in c.items()))
