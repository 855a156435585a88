for rep in 1 2; do
timeout -k 10 150 python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('gate after shared stage', round(d['ms_per_step'],3))"
BASD_EXP_GATE_AFTER=1 timeout -k 10 150 python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('gate after tail', round(d['ms_per_step'],3))"
done
