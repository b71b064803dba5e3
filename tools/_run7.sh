set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
export TMPDIR=/tmp
python tools/gc_stats.py > gpurun_out/r3g/gc_stats.txt 2>&1; tail -5 gpurun_out/r3g/gc_stats.txt
bash tools/profile_round.sh r03 > gpurun_out/r3g/profile_round.log 2>&1
bash tools/power_sample.sh gpurun_out/prof_r03/power_under_pipeline.txt python bench.py --steps 12000 --warmup 10 --no-cpu-baseline --no-roofline --no-stage-times --traffic off
for f in bench_driver_cmd bench_100steps bench_100steps_b bench_bf16x6 bench_fp32_mfma bench_sparse_heads bench_c3_shard_batch4 bench_c1 bench_c4 bench_c5; do python - $f <<'PY'
import json,sys
try:
  d=json.loads(open('gpurun_out/prof_r03/%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
  r=d.get('roofline',{})
  print(sys.argv[1], d['value'], d['ms_per_step'], d.get('serial_depth1',{}).get('images_per_sec'), r.get('achieved'), r.get('frac'), r.get('in_step',{}).get('gemm_ms_per_step'), d['config']['workload'][:40])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
cat gpurun_out/prof_r03/power_under_pipeline.txt | head -14
