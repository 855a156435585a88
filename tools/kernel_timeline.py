"""Kernel timeline of ONE step from a rocprofv3 --kernel-trace csv: start / end (us since the step's first kernel), queue
and kernel, in start order.  usage: kernel_timeline.py <kernel_trace.csv> [marker substring, default cross_entropy]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "cross_entropy"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
s, e = idx[-2], idx[-1]
t0 = int(rows[s]["Start_Timestamp"])
for r in rows[s:e]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("basd::", "")[:48]
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%8.1f %8.1f %7.1f  q%-3s %s" % (a, b, b - a, r.get("Queue_Id", "?"), name))
