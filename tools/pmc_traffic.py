"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; counter_collection csv) into bytes per launch.
usage: pmc_traffic.py <fetch_csv> <write_csv> <cfg> <out_json>
gfx950: FETCH_SIZE under-reports wide coalesced reads by exactly 2 (MI355X_MICROARCH.md, HBM section); both
counters are in KiB-equivalent units of 1024 bytes?  No: rocprofv3 reports them in KILOBYTES (derived metric)."""
import csv, json, sys, collections

def per_kernel(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        key = "tridiag_kernel" if "tridiag_kernel" in name else "syrk_tn_kernel" if "syrk_tn_kernel" in name else None
        if key is None:
            continue
        key = (key, int(r["Grid_Size"]))
        tot[key] += float(r["Counter_Value"])
        cnt[key] += 1
    # the entry points are called for the teacher and for the student matrices: keep the larger grid (student)
    best = {}
    for (k, grid) in tot:
        if k not in best or grid > best[k]:
            best[k] = grid
    return {k: tot[(k, g)] / cnt[(k, g)] for k, g in best.items()}, {k: cnt[(k, g)] for k, g in best.items()}

fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
cfg, out = sys.argv[3], sys.argv[4]
res = {}
for k in fetch:
    res[k] = int(2 * fetch[k] * 1024 + write.get(k, 0.0) * 1024)     # bytes per launch (mean over launches)
try:
    allc = json.load(open(out))
except (OSError, ValueError):
    allc = {}
allc[cfg] = res
allc.setdefault("_how", "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 10 "
                "--warmup 3 --no-cpu-baseline`; bytes = 2 x FETCH_SIZE KB (gfx950 correction) + WRITE_SIZE KB, mean per launch")
json.dump(allc, open(out, "w"), indent=1)
print(res, nf, nw)
