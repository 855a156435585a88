"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; counter_collection csv) into bytes per launch.
usage: pmc_traffic.py <fetch_csv> <write_csv> <cfg> <out_json>
gfx950: FETCH_SIZE under-reports wide coalesced reads by exactly 2 (MI355X_MICROARCH.md, HBM section); rocprofv3 reports
both derived counters in kilobytes.  Kernels are keyed by short name; where an entry point is called with two grids per
step (teacher / student side) the LARGER grid's launches are kept."""
import csv, json, sys, collections

KEYS = {"tridiag_packed_kernel": "tridiag_packed_kernel", "tridiag_kernel": "tridiag_kernel",
        "tridiag_tail2_kernel": "tridiag_tail_kernel", "syrk_tn_kernel": "syrk_tn_kernel",
        "syrk_tn_split_kernel": "syrk_tn_kernel", "gemm_nt_split_kernel<float, false>": "gemm_nt_kernel",
        "jacobi_oe_kernel": "jacobi_lds_kernel",
        "colsum_partial_vec_kernel": "colsum_partial_vec_kernel", "gemm_nt_kernel<float, false>": "gemm_nt_kernel",
        "jacobi_lds_kernel<16, 16, 4>": "jacobi_lds_kernel", "student_project_v4_kernel": "student_project_v4_kernel",
        "student_grad_fused_kernel": "student_grad_fused_kernel"}


def per_kernel(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        key = next((v for k, v in KEYS.items() if k in r["Kernel_Name"]), None)
        if key is None:
            continue
        key = (key, int(r["Grid_Size"]))
        tot[key] += float(r["Counter_Value"])
        cnt[key] += 1
    best = {}
    for (k, grid) in tot:
        if k not in best or grid > best[k]:
            best[k] = grid
    return {k: tot[(k, g)] / cnt[(k, g)] for k, g in best.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
cfg, out = sys.argv[3], sys.argv[4]
res = {k: int(2 * fetch[k] * 1024 + write.get(k, 0.0) * 1024) for k in fetch}    # bytes per launch (mean over launches)
# one factorisation: the packed kernel where it applies (orders 257..384), else the two kernels of the two-stage path
res["tridiag"] = res.get("tridiag_packed_kernel") or (res.get("tridiag_kernel", 0) + res.get("tridiag_tail_kernel", 0))
try:
    allc = json.load(open(out))
except (OSError, ValueError):
    allc = {}
allc[cfg] = res
allc["_how"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 10 "
                "--warmup 3 --no-cpu-baseline [--config ...]`; bytes = 2 x FETCH_SIZE KB (gfx950 correction) + WRITE_SIZE KB, "
                "mean per launch")
json.dump(allc, open(out, "w"), indent=1)
print(res)
