#!/bin/bash
# ransac_hypotheses with and without the exact pruning (EPOS_FIT_PRUNE=0|1), on the planted
# workload at 30 / 50 / 70 % outlier pixels and on the default (random heads) workload:
# rocprofv3 kernel stats of a short serial run each.   bash tools/ab_prune.sh > out.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
for wl in "--planted-poses --planted-outliers 0.3" "--planted-poses --planted-outliers 0.5" "--planted-poses --planted-outliers 0.7" "" "--planted-poses --planted-outliers 0.5 --height 540 --width 720 --num-objs 30 --instances 4"; do
  for prune in 0 1; do
    d=$(mktemp -d /tmp/prune_XXXX)
    (cd /tmp && EPOS_FIT_PRUNE=$prune rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $ROOT/bench.py $wl --pipeline-depth 1 --timed-repeats 1 --steps 30 --no-cpu-baseline --traffic off --no-roofline --no-stage-times > $d/bench.json 2> $d/err.txt)
    python - "$d" "$wl" "$prune" <<'PY'
import csv, glob, json, sys
d, wl, prune = sys.argv[1:4]
try:
  v = json.loads(open(d + '/bench.json').read().strip().split('\n')[-1])
  ips, ok = v['value'], v.get('planted', {}).get('ok')
except Exception as e:
  ips, ok = None, 'bench failed: %s' % open(d + '/err.txt').read()[-300:]
out = []
for f in glob.glob(d + '/**/*kernel_stats.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    for k in ('ransac_hypotheses', 'ransac_select_lo', 'ransac_refit_accept'):
      if k in r['Name']:
        out.append('%s %d calls avg %.1f us' % (k, int(r['Calls']), float(r['AverageNs']) / 1e3))
print('workload [%s] EPOS_FIT_PRUNE=%s: serial %s images/s, planted ok=%s; %s' % (wl or 'default (random heads)', prune, ips, ok, '; '.join(sorted(out))))
PY
    rm -rf $d
  done
done
