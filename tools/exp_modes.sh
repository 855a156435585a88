for rep in 1 2; do for tg in 1 0; do
BASD_TAIL_GATE=$tg timeout -k 10 150 python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('tail_gate=$tg', round(d['ms_per_step'],3), d['loss'])"
done; done
BASD_TAIL_GATE=1 timeout -k 10 200 python tools/step_clock.py 2>&1 | grep -v amdgpu | egrep "GPU|ranks_read|step_out|fwd_out"
