timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k tridiag 2>&1 | tail -2
timeout -k 10 200 python tools/tridiag_bench.py 2>&1 | grep -v amdgpu | grep "tail=[01]"
for rep in 1 2; do
timeout -k 10 150 python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['ms_per_step'],3), d['loss'])"
done
timeout -k 10 200 python tools/step_clock.py 2>&1 | grep -v amdgpu | egrep "GPU|ranks_read|step_out"
