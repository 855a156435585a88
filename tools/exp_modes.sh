for rep in 1 2; do for r in 0 1 2; do
BASD_CU_RESERVE=$r timeout -k 10 150 python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('reserve=$r', round(d['ms_per_step'],3), d['loss'])"
done; done
BASD_CU_RESERVE=1 timeout -k 10 200 python tools/step_clock.py 2>&1 | grep -v amdgpu | egrep "GPU|ranks_read|step_out|fwd_out|Error|error"
BASD_CU_RESERVE=1 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
