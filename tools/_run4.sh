set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
export TMPDIR=/tmp
python tools/h2_race.py > gpurun_out/r3d/h2_race.txt 2>&1; tail -12 gpurun_out/r3d/h2_race.txt
timeout 600 python -m pytest tests/test_gpu_pipeline.py -x -q -k "concurrent" 2>&1 | tail -5 > gpurun_out/r3d/det_h2.log; cat gpurun_out/r3d/det_h2.log
EPOS_GEMM_H2=0 timeout 600 python -m pytest tests/test_gpu_pipeline.py -x -q -k "concurrent" 2>&1 | tail -5 > gpurun_out/r3d/det_split.log; cat gpurun_out/r3d/det_split.log
timeout 900 python -m pytest tests/test_gpu_corresp_fit.py tests/test_gpu_boundary.py -x -q 2>&1 | tail -6 > gpurun_out/r3d/fit_tests.log; cat gpurun_out/r3d/fit_tests.log
(cd /tmp && rm -rf /tmp/prof_a && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --pipeline-depth 1 --no-cpu-baseline --no-roofline --no-stage-times --traffic off > /dev/null 2>&1; cp $(find /tmp/prof_a -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r3d/kernel_stats_depth1.csv)
grep -h "ransac" gpurun_out/r3d/kernel_stats_depth1.csv | cut -c1-60,200-400
python tools/bench_gemm_h2_abl.py > gpurun_out/r3d/h2_ablations.txt 2>&1; cat gpurun_out/r3d/h2_ablations.txt
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --traffic off > gpurun_out/r3d/bench.json 2> gpurun_out/r3d/bench.err; tail -c 1500 gpurun_out/r3d/bench.json
