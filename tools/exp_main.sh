timeout -k 10 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
timeout -k 10 200 python tools/step_clock.py 2>&1 | grep -v amdgpu
for i in 1 2; do timeout -k 10 150 python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench ms/step', round(d['ms_per_step'],3), 'loss', d['loss'])"; done
