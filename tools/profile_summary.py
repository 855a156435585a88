"""Write profiles/<tag>_bench_<cfg>_summary.md (+ copy the kernel-stats csv) from three rocprofv3 passes of
`python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline`:
    --kernel-trace --stats   ->  <stats_dir>/*kernel_stats.csv
    --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE  ->  *counter_collection.csv
usage: profile_summary.py <tag> <cfg> <stats_csv> <fetch_csv> <write_csv> [bench_json_under_rocprof]"""
import csv, sys, os, shutil, collections, json

tag, cfg, stats_csv, fetch_csv, write_csv = sys.argv[1:6]
under = sys.argv[6] if len(sys.argv) > 6 else None
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(root, "profiles")


def short(name):
    name = name.replace("void ", "")
    cut = name.find("(")
    return (name[:cut] if cut > 0 else name)[:72]


def pmc(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = short(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}


fetch, write = pmc(fetch_csv, "FETCH_SIZE"), pmc(write_csv, "WRITE_SIZE")
rows = list(csv.DictReader(open(stats_csv)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
shutil.copy(stats_csv, os.path.join(out_dir, f"{tag}_bench_{cfg}_kernel_stats.csv"))
if under:
    shutil.copy(under, os.path.join(out_dir, f"{tag}_bench_{cfg}_under_rocprof.json"))
lines = [
    f"# rocprofv3 summary, {tag} -- `python bench.py --config {cfg} --steps 10 --warmup 3 --no-cpu-baseline` (1 x MI355X)",
    "",
    f"Source: `rocprofv3 --kernel-trace --stats` (`{tag}_bench_{cfg}_kernel_stats.csv` next to this file, 13 steps);",
    "HBM-side traffic from separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, corrected as MI355X_MICROARCH.md",
    "prescribes for gfx950 (bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024; FETCH_SIZE counts 64 B of each 128 B",
    "request; Infinity-Cache hits are included in these counters).  Mean per launch over all launches of a kernel",
    f"(the teacher-side and student-side calls of one kernel are averaged together here; `{tag}_traffic.json` holds the",
    "student-side launch alone for the kernels `bench.py` reports a roofline for).  Durations are in-step: up to four",
    "streams share the chip, so a kernel's time here includes what the others cost it.",
    "",
    "| kernel | calls | total ms | avg us | % | HBM-side KB/launch (PMC) |",
    "|---|---|---|---|---|---|",
]
for r in rows[:32]:
    k = short(r["Name"])
    kb = 2 * fetch.get(k, 0.0) + write.get(k, 0.0)
    lines.append(f"| `{k}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | "
                 f"{float(r['Percentage']):.1f} | {kb:.0f} |")
open(os.path.join(out_dir, f"{tag}_bench_{cfg}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:24]))
