mkdir -p gpurun_out/r2l
export TMPDIR=/tmp
for lanes in 16 4; do
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d /tmp/pmc_a$lanes -o p --output-format csv -- python3 tools/jacobi_pmc.py $lanes > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAVES SQ_INSTS_SMEM -d /tmp/pmc_b$lanes -o p --output-format csv -- python3 tools/jacobi_pmc.py $lanes > /dev/null 2>&1
for x in a b; do f=$(find /tmp/pmc_$x$lanes -name "*counter_collection.csv" | head -1); python3 - "$f" "$lanes$x" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "jacobi_lds" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: sum(v) / len(v) for k, v in acc.items()})
PY
done; done
