set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_corresp_fit.py tests/test_gpu_boundary.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -6 > gpurun_out/r3h/fit_tests.log; cat gpurun_out/r3h/fit_tests.log
(cd /tmp && rm -rf /tmp/prof_a && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --pipeline-depth 1 --no-cpu-baseline --no-roofline --no-stage-times --traffic off > /dev/null 2>&1; cp $(find /tmp/prof_a -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r3h/kernel_stats_depth1_noroofline.csv)
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r3h/kernel_stats_depth1_noroofline.csv')):
  if 'ransac' in r['Name'] or 'corr_' in r['Name']: print(r['Name'].split('(')[0][-32:], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --traffic off > gpurun_out/r3h/bench.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3h/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['serial_depth1'])
PY
