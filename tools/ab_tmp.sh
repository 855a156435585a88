run() { python bench.py --config $2 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', round(d['ms_per_step'],3), d['loss'])"; }
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/t14.log 2>&1; tail -5 gpurun_out/t14.log
for c in cfg4 cfg2; do
BASD_GEMM_SPLIT=1 run split $c
done
