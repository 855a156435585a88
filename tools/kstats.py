"""Per-step kernel time table from a rocprofv3 kernel_stats csv.  usage: kstats.py <csv> <steps incl. warm-up> [filter ...]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
filt = sys.argv[3:]
tot = 0.0
for r in rows:
    name = r["Name"].split("(")[0].replace("void ", "").replace("basd::", "")
    t = float(r["TotalDurationNs"]) / steps / 1e6
    tot += t
    if filt and not any(f in name for f in filt):
        continue
    if not filt and t < 0.004:
        continue
    print("%-52s %5d  per-step %7.3f ms  avg %8.1f us" % (name[:52], int(r["Calls"]), t, float(r["AverageNs"]) / 1e3))
print("sum of kernel time per step: %.3f ms" % tot)
