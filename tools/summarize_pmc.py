#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc counter_collection CSVs (gpurun_out/<dir>/*/ *_counter_collection.csv) into one
JSON: mean counter value per kernel over its last 5 dispatches.  usage: summarize_pmc.py out.json dir1 dir2 ..."""
import collections
import csv
import glob
import json
import sys

out, dirs = sys.argv[1], sys.argv[2:]
res = collections.defaultdict(dict)
for d in dirs:
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            vv = v[-5:]
            res[k][c] = sum(vv) / len(vv)
            res[k]["dispatches_seen"] = len(v)
json.dump({k: v for k, v in sorted(res.items()) if not k.startswith("void at::") and "rocclr" not in k}, open(out, "w"), indent=1)
print("wrote", out, len(res), "kernels")
