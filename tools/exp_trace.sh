mkdir -p gpurun_out/r2h
export TMPDIR=/tmp
for mode in sync deferred; do
  BASD_RANK_READBACK=$mode rocprofv3 --kernel-trace -d /tmp/tr_$mode -o s -- python3 bench.py --steps 10 --warmup 8 --no-cpu-baseline > gpurun_out/r2h/bench_$mode.json 2> gpurun_out/r2h/bench_$mode.err
  python3 -c "import json; d=json.load(open('gpurun_out/r2h/bench_$mode.json')); print('$mode', d['ms_per_step'])"
  python3 tools/trace_timeline.py /tmp/tr_$mode/s_results.db token_weights_kernel 2 2 > gpurun_out/r2h/timeline_$mode.txt 2>&1
done
