set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
export TMPDIR=/tmp
python tools/diag_concurrent.py 2>&1 | tail -4 | cut -c1-300 > gpurun_out/r3f/diag_concurrent.txt; cat gpurun_out/r3f/diag_concurrent.txt
timeout 900 python -m pytest tests/test_gpu_h2.py -x -q 2>&1 | tail -8 > gpurun_out/r3f/h2_tests.log; cat gpurun_out/r3f/h2_tests.log
for v in 1 0; do
  EPOS_H2_PRESPLIT=$v python bench.py --steps 60 --warmup 5 --no-cpu-baseline --traffic off > gpurun_out/r3f/bench_presplit$v.json 2> gpurun_out/r3f/bench_presplit$v.err
done
EPOS_H2_PRESPLIT=1 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --traffic off > gpurun_out/r3f/bench_presplit1_b.json 2>/dev/null
python - <<'PY'
import json
for f in ('bench_presplit1.json','bench_presplit0.json','bench_presplit1_b.json'):
  d=json.loads(open('gpurun_out/r3f/'+f).read().strip().splitlines()[-1])
  print(f, d['value'], d['ms_per_step'], d['serial_depth1']['images_per_sec'], d['roofline']['in_step']['gemm_ms_per_step'], d['roofline']['in_step']['depthwise_ms_per_step'], d['roofline']['in_step']['rest_ms_per_step'], d['roofline']['achieved'])
PY
(cd /tmp && rm -rf /tmp/pmc1 && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc1 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --pipeline-depth 1 --no-cpu-baseline --no-roofline --no-stage-times --traffic off > /dev/null 2>&1; f=$(find /tmp/pmc1 -name "*counter_collection.csv" | head -1); head -1 $f > $GRAFT_REPO_ROOT/gpurun_out/r3f/pmc_fit.csv; grep -E "ransac|pointwise_gemm_h2" $f | head -400 >> $GRAFT_REPO_ROOT/gpurun_out/r3f/pmc_fit.csv)
wc -l gpurun_out/r3f/pmc_fit.csv
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3f/all_gpu_tests.log
cat gpurun_out/r3f/all_gpu_tests.log
