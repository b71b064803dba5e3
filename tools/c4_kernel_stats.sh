#!/bin/bash
# rocprofv3 kernel statistics of the C4 configuration at pipeline depth 1 (the fitting stage's
# kernels per call): bash tools/c4_kernel_stats.sh <tag>   [EPOS_HIP_LIB=<variant> in the env]
T=${1:-c4}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4/kt_$T -- python bench.py --steps 20 --warmup 5 --height 540 --width 720 --num-objs 30 --objs-per-image 8 --instances 2 --pipeline-depth 1 --no-cpu-baseline --traffic static --no-stage-times > gpurun_out/c4/log_$T.txt 2>&1
f=$(find gpurun_out/c4/kt_$T -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/c4/kernel_stats_$T.csv; rm -rf gpurun_out/c4/kt_$T
grep -E "pearl|ransac" gpurun_out/c4/kernel_stats_$T.csv | awk -F'","' '{split($1,a,"::"); n=a[length(a)]; sub(/\(.*/,"",n); printf "%-24s calls %6s avg %8.1f us\n", n, $2, $4/1000}' | sed 's/"//g'
