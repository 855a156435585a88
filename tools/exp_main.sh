mkdir -p gpurun_out/r2m
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "jacobi" 2>&1 | tail -3
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do timeout -k 10 150 python tools/host_timeline.py 2>&1 | grep -v amdgpu | egrep "ms/step|ranks_read|procrustes_q|step_out|teacher chain"; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/st -o s --output-format csv -- python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline > gpurun_out/r2m/bench_prof.json 2>/dev/null
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/st/**/*kernel_stats.csv', recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:26]:
    print(f"{float(r['TotalDurationNs'])/14e3:9.1f} us/step  x{int(r['Calls'])/14:4.1f}  avg {float(r['AverageNs'])/1e3:8.1f}  {r['Name'][:70]}")
PY
