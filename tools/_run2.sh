set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_corresp_fit.py tests/test_gpu_h2.py -x -q 2>&1 | tail -15 > gpurun_out/r3b/fit_tests.log
cat gpurun_out/r3b/fit_tests.log
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --traffic off > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3b/all_gpu_tests.log
cat gpurun_out/r3b/all_gpu_tests.log
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_d1 -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --pipeline-depth 1 --no-cpu-baseline --no-roofline --no-stage-times --traffic off > /dev/null 2>&1; cp $(find /tmp/prof_d1 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r3b/kernel_stats_depth1.csv)
head -30 gpurun_out/r3b/kernel_stats_depth1.csv
