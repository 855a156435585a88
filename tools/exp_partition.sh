mkdir -p gpurun_out/r2g
run() { echo "== $1 partition=$2"; BASD_CU_PARTITION=$2 BASD_RANK_READBACK=$1 timeout -k 10 150 python tools/host_timeline.py 2>&1 | grep -v amdgpu.ids | egrep "ms/step|ranks_read|procrustes_queued|step_out|Error|error" ; }
for p in 0 4,3 4,4 6,4 3,2 8,4 0; do run sync $p; done
for p in 0 4,3 6,4 8,4; do run deferred $p; done
