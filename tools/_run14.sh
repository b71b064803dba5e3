set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_corresp_fit.py tests/test_gpu_fit_lists.py tests/test_gpu_boundary.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -5 > gpurun_out/r4b/tests.log; cat gpurun_out/r4b/tests.log
python tools/fit_trace.py > gpurun_out/r4b/fit_trace.txt 2>&1; cat gpurun_out/r4b/fit_trace.txt | tail -30
cd /tmp
rm -rf /tmp/prof_a
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -- python /root/repo/bench.py --steps 30 --warmup 5 --pipeline-depth 1 --no-cpu-baseline --no-roofline --no-stage-times --traffic off > /dev/null 2>&1
f=$(find /tmp/prof_a -name '*kernel_stats.csv' | head -1)
python - $f <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
  if 'ransac' in r['Name'] or 'pearl' in r['Name']:
    print(r['Name'][:60], r['Calls'], '%.1f'%(float(r['AverageNs'])/1e3))
P
cd /root/repo
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --traffic off > gpurun_out/r4b/bench.json 2>gpurun_out/r4b/bench.err
python - <<'P'
import json
for n in ('bench',):
  j=json.loads(open('/root/repo/gpurun_out/r4b/%s.json'%n).read().strip().splitlines()[-1])
  print(n, j['value'], j['ms_per_step'], j.get('serial_depth1',{}).get('stage_ms'))
P
