"""Diagnostic: per-kernel timeline of ONE bench step from a rocprofv3 kernel trace (csv path as argv[1])."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# take the last step: find last occurrence of token_weights kernel start
idx = [i for i, r in enumerate(rows) if "teacher_center" in r["Kernel_Name"]]
start = idx[-1] - 40
t0 = None
for r in rows[max(0, start):]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None:
        t0 = s
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{r['Queue_Id']:>3s} {r['Kernel_Name'][:70]}")
