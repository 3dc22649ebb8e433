#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc counter_collection CSVs (gpurun_out/<dir>/*/ *_counter_collection.csv) into one
JSON: mean counter value per kernel over its last 5 dispatches, stamped with the workload size (PBR_PROFILE_PIXELS,
default 3840*2160) and the sha1 of the shade sources (bench.py ignores a summary whose stamp differs from the tree's).
usage: summarize_pmc.py out.json dir1 dir2 ..."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

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
from bench import src_stamp  # noqa: E402
doc = {k: v for k, v in sorted(res.items()) if not k.startswith("void at::") and "rocclr" not in k}
doc["_workload_pixels"] = int(os.environ.get("PBR_PROFILE_PIXELS", 3840 * 2160))
doc["_src_stamp"] = src_stamp()
json.dump(doc, open(out, "w"), indent=1)
print("wrote", out, len(res), "kernels")
