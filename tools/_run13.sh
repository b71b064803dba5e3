set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3m
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_corresp_fit.py tests/test_gpu_fit_lists.py tests/test_gpu_boundary.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -12 > gpurun_out/r3m/tests.log; cat gpurun_out/r3m/tests.log
cd /tmp
for mode in 1 0; do
rm -rf /tmp/prof_a
EPOS_FIT_SCAN=$mode rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -- python /root/repo/bench.py --steps 30 --warmup 5 --pipeline-depth 1 --no-cpu-baseline --no-roofline --no-stage-times --traffic off > /dev/null 2>&1
f=$(find /tmp/prof_a -name '*kernel_stats.csv' | head -1)
cp $f /root/repo/gpurun_out/r3m/kernel_stats_depth1_scan$mode.csv
python - $f <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
  if 'ransac' in r['Name'] or 'pearl' in r['Name']:
    print(r['Name'][:60], r['Calls'], '%.1f'%(float(r['AverageNs'])/1e3))
P
done
cd /root/repo
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --traffic off > gpurun_out/r3m/bench.json 2>gpurun_out/r3m/bench.err
EPOS_FIT_SCAN=0 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --traffic off > gpurun_out/r3m/bench_scan0.json 2>/dev/null
python bench.py --steps 40 --warmup 5 --height 540 --width 720 --num-objs 30 --objs-per-image 8 --instances 2 --pipeline-depth 3 --no-cpu-baseline --traffic off > gpurun_out/r3m/bench_c4.json 2>/dev/null
python - <<'P'
import json
for n in ('bench','bench_scan0','bench_c4'):
  try:
    j=json.loads(open('/root/repo/gpurun_out/r3m/%s.json'%n).read().strip().splitlines()[-1])
    print(n, j['value'], j['ms_per_step'], j.get('serial_depth1',{}).get('images_per_s'), j.get('serial_depth1',{}).get('stage_ms'))
  except Exception as e: print(n,'ERR',e)
P
