mkdir -p gpurun_out/r2k
for rep in 1 2; do for g in 0 1; do
  echo "== gate=$g rep=$rep"
  BASD_STUDENT_GATE=$g timeout -k 10 150 python tools/host_timeline.py 2>&1 | grep -v amdgpu | egrep "ms/step|ranks_read|procrustes_q|step_out|teacher chain"
done; done
