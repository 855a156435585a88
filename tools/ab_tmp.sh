run() { python bench.py --config cfg2 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3))"; }
for i in 1 2; do
run base
BASD_CHAIN_SLOTS=4 run slots4
BASD_STUDENT_LOW_PRIORITY=0 run student_normal
BASD_CHAIN_PRIORITY=0 run chain_normal
BASD_CHAIN_PRIORITY=1 run chain_student_high
BASD_TAIL_LOW_PRIORITY=1 run tail_low
BASD_PROCRUSTES_FIRST=1 run proc_first
done
