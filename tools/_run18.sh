cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3x
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_corresp_fit.py tests/test_gpu_fit_lists.py tests/test_gpu_boundary.py -x -q 2>&1 | tail -3
cd /tmp
rm -rf /tmp/prof_a
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -- python /root/repo/bench.py --steps 12 --warmup 3 --height 540 --width 720 --num-objs 30 --objs-per-image 8 --instances 2 --pipeline-depth 1 --no-cpu-baseline --no-roofline --no-stage-times --traffic off > /dev/null 2>&1
f=$(find /tmp/prof_a -name '*kernel_stats.csv' | head -1)
cp $f /root/repo/gpurun_out/r3x/kernel_stats_c4_depth1.csv
python - $f <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
  if 'ransac' in r['Name'] or 'pearl' in r['Name']:
    print('%-60s %6s %8.1f %8.1f'%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3/15))
P
cd /root/repo
python bench.py --steps 40 --warmup 5 --height 540 --width 720 --num-objs 30 --objs-per-image 8 --instances 2 --pipeline-depth 3 --no-cpu-baseline --traffic static 2>/dev/null | tail -1 > gpurun_out/r3x/bench_c4.json
python - <<'P'
import json
for n in ('bench_c4',):
  j=json.loads(open('/root/repo/gpurun_out/r3x/%s.json'%n).read().strip().splitlines()[-1])
  print(n, j['value'], j['ms_per_step'], j.get('serial_depth1'))
P
