# Run on the GPU box: rocprofv3 kernel stats (+ PMC traffic for cfg2), the bench lines of every config, the steady-state
# step clock, the secondary workload with a rank-160 teacher, and one full-batch CPU-oracle record -> gpurun_out/r03/
set -u
OUT=gpurun_out/r03
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline"
for cfg in cfg2 cfg1 cfg5 cfg4; do
  rocprofv3 --kernel-trace --stats -d /tmp/st_$cfg -o s --output-format csv -- $BENCH --config $cfg > $OUT/bench_${cfg}_under_rocprof.json 2> $OUT/bench_${cfg}_rocprof.err
  cp $(find /tmp/st_$cfg -name "*kernel_stats.csv" | head -1) $OUT/${cfg}_kernel_stats.csv
  echo "$cfg under rocprof: $(python3 -c "import json;print(json.load(open('$OUT/bench_${cfg}_under_rocprof.json'))['ms_per_step'])")"
done
rocprofv3 --kernel-trace --stats -d /tmp/st_k160 -o s --output-format csv -- $BENCH --teacher-rank 160 > $OUT/bench_cfg2_rank160_under_rocprof.json 2> $OUT/bench_cfg2_rank160_rocprof.err
cp $(find /tmp/st_k160 -name "*kernel_stats.csv" | head -1) $OUT/cfg2_rank160_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p --output-format csv -- $BENCH --config cfg2 > /dev/null 2> $OUT/pmc_$c.err
  cp $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) $OUT/cfg2_$c.csv
done
python3 tools/pmc_traffic.py $OUT/cfg2_FETCH_SIZE.csv $OUT/cfg2_WRITE_SIZE.csv cfg2 $OUT/r03_traffic.json
# un-profiled bench lines (the numbers to quote); cfg2 with the CPU baseline leg
python3 bench.py > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
for cfg in cfg1 cfg5 cfg4; do python3 bench.py --config $cfg --no-cpu-baseline > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err; done
python3 bench.py --teacher-rank 160 --no-cpu-baseline > $OUT/bench_cfg2_rank160.json 2> $OUT/bench_cfg2_rank160.err
for cfg in cfg2 cfg1 cfg5 cfg4 cfg2_rank160; do python3 -c "import json;d=json.load(open('$OUT/bench_$cfg.json'));print('$cfg', round(d['ms_per_step'],3),'ms/step', round(d['value'],1),'img/s loss',d['loss'], {k: round(d[k]['frac'],4) for k in ('roofline','roofline_mfma','roofline_hbm_stream')})"; done
python3 tools/step_clock.py cfg2 > $OUT/step_clock_cfg2.txt 2>&1
BASD_TRIDIAG_CLOCKS=1 python3 tools/step_clock.py cfg2 2>&1 | grep -i "factorisation kernel" > $OUT/packed_kernel_in_step.txt
python3 tools/tridiag_bench.py > $OUT/tridiag_bench.txt 2>&1
# the headline workload at its full batch on the host cores: one warm-up + one timed step of the CPU oracle (minutes)
python3 bench.py --cpu-baseline-only --cpu-sample-batch 256 --cpu-repeats 1 > $OUT/cpu_oracle_cfg2_full_batch.json 2> $OUT/cpu_oracle.err
# keep the merge small: the PMC csv files are large
gzip -f $OUT/cfg2_FETCH_SIZE.csv $OUT/cfg2_WRITE_SIZE.csv
ls -la $OUT
