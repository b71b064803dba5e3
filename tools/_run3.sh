set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_corresp_fit.py -x -q 2>&1 | tail -15 > gpurun_out/r3c/fit_tests.log
cat gpurun_out/r3c/fit_tests.log
for v in "" gcnoloop; do
  if [ -n "$v" ]; then export EPOS_HIP_LIB=$GRAFT_REPO_ROOT/epos_amd/lib/libepos_hip_$v.so; else unset EPOS_HIP_LIB; fi
  (cd /tmp && rm -rf /tmp/prof_$v && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --pipeline-depth 1 --no-cpu-baseline --no-roofline --no-stage-times --traffic off > /dev/null 2>&1; cp $(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r3c/kernel_stats_depth1_$v.csv)
done
unset EPOS_HIP_LIB
grep -h "ransac" gpurun_out/r3c/kernel_stats_depth1_*.csv | cut -c1-60,200-400
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3c/all_gpu_tests.log
cat gpurun_out/r3c/all_gpu_tests.log
