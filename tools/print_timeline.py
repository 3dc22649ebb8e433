"""Prints the last frame (from the last clustered launch on) of a rocprofv3 kernel_trace.csv: duration and gap per launch."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_cluster_cull<true>" in r["Kernel_Name"] or r["Kernel_Name"].startswith("k_cluster_build")]
s = idx[-1]
prev_end = None
t0 = int(rows[s]["Start_Timestamp"])
for r in rows[s:]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
    print(f"{name:44s} grid {r['Grid_Size_X']:>8}x{r['Grid_Size_Y']:>5} wg {r['Workgroup_Size_X']:>4} dur {(en - st) / 1e3:7.1f} us gap {gap:5.1f} t={(st - t0) / 1e3:7.1f}")
    prev_end = en
print(f"frame span {(prev_end - t0) / 1e3:.1f} us")
