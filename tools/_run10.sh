set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_corresp_fit.py tests/test_gpu_fit_lists.py tests/test_gpu_boundary.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -12 > gpurun_out/r3j/tests.log; cat gpurun_out/r3j/tests.log
(cd /tmp && rm -rf /tmp/prof_a && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --pipeline-depth 1 --no-cpu-baseline --no-roofline --no-stage-times --traffic off > /dev/null 2>&1; cp $(find /tmp/prof_a -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r3j/kernel_stats_depth1_noroofline.csv)
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r3j/kernel_stats_depth1_noroofline.csv')):
  if 'ransac' in r['Name']: print(r['Name'].split('(')[0][-36:], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --traffic off > gpurun_out/r3j/bench.json 2>/dev/null
python bench.py --steps 40 --warmup 5 --height 540 --width 720 --num-objs 30 --objs-per-image 8 --instances 2 --pipeline-depth 3 --no-cpu-baseline --traffic off > gpurun_out/r3j/bench_c4.json 2>/dev/null
EPOS_FIT_NB_LISTS=0 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --traffic off > gpurun_out/r3j/bench_nolists.json 2>/dev/null
python - <<'PY'
import json
for f in ('bench','bench_nolists','bench_c4'):
  d=json.loads(open('gpurun_out/r3j/%s.json'%f).read().strip().splitlines()[-1])
  print(f, d['value'], d['ms_per_step'], d['serial_depth1']['images_per_sec'], d['serial_depth1']['stage_ms'])
PY
