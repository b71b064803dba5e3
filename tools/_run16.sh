set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r4c/all_gpu_tests.log; cat gpurun_out/r4c/all_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3
bash tools/profile_round.sh r03d > gpurun_out/r4c/profile_round.log 2>&1
python tools/fit_trace.py > gpurun_out/prof_r03d/fit_trace.txt 2>&1
tail -5 gpurun_out/prof_r03d/fit_trace.txt
