mkdir -p gpurun_out/r06d; O=gpurun_out/r06d
(timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_net.py tests/test_gpu_configs.py tests/test_gpu_corresp_fit.py tests/test_gpu_pipeline.py tests/test_gpu_fit_lists.py -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log)
COMMON="--steps 60 --warmup 8 --no-cpu-baseline --traffic static" REPS=3 bash tools/ab_env.sh "EPOS_PAD_ROWS=0" "EPOS_PAD_ROWS=1" > $O/ab_pad_rows.txt 2>&1; cat $O/ab_pad_rows.txt
COMMON="--steps 40 --warmup 6 --no-cpu-baseline --traffic static --batch-per-gpu 4" REPS=2 bash tools/ab_env.sh "EPOS_PAD_ROWS=0" "EPOS_PAD_ROWS=1" > $O/ab_pad_rows_c3.txt 2>&1; cat $O/ab_pad_rows_c3.txt
bash tools/pmc_dw_traffic.sh > $O/pmc_dw_traffic.txt 2>&1; cat $O/pmc_dw_traffic.txt
for L in "" "--pad32"; do echo "== bench_dw --h2 $L"; python tools/bench_dw.py --h2 $L 2>&1 | head -3; done > $O/bench_dw_pad.txt; cat $O/bench_dw_pad.txt
cd /tmp; export TMPDIR=/tmp; ROOT=$GRAFT_REPO_ROOT
for wl in "--planted-poses --planted-outliers 0.5" ""; do for prune in 0 1 0 1; do
  d=$(mktemp -d /tmp/prune_XXXX)
  EPOS_FIT_PRUNE=$prune rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $ROOT/bench.py $wl --pipeline-depth 1 --timed-repeats 1 --steps 30 --no-cpu-baseline --traffic off --no-roofline --no-stage-times > $d/bench.json 2> $d/err.txt
  python - "$d" "$wl" "$prune" <<'PY'
import csv, glob, json, sys
d, wl, prune = sys.argv[1:4]
out = []
for f in glob.glob(d + '/**/*kernel_stats.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    if 'ransac_hypotheses' in r['Name']:
      out.append('ransac_hypotheses %d calls avg %.1f us' % (int(r['Calls']), float(r['AverageNs']) / 1e3))
print('[%s] EPOS_FIT_PRUNE=%s (prefetched bound): %s' % (wl or 'default', prune, '; '.join(out)))
PY
  rm -rf $d
done; done 2>&1 | tee $ROOT/$O/ransac_pruning_ab_v2.txt
cd $ROOT
timeout 900 bash tools/infer_diag.sh $O/infer_diag
