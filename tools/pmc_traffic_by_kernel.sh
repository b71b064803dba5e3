#!/bin/bash
# Fabric-side bytes per image BY KERNEL (round 6: "removing bytes pays" -- where are they?):
# FETCH_SIZE (x2 on gfx950) and WRITE_SIZE in separate passes over a short serial bench run.
#   bash tools/pmc_traffic_by_kernel.sh > gpurun_out/traffic_by_kernel.txt
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
D=$(mktemp -d /tmp/pmck_XXXX)
for c in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $D/$c
  rocprofv3 --pmc $c --output-format csv -d $D/$c -- python bench.py --steps 6 --warmup 2 --timed-repeats 1 --pipeline-depth 1 --no-cpu-baseline --no-roofline --no-stage-times --traffic off > $D/$c.log 2>&1
done
python - "$D" <<'PY'
import csv, glob, re, sys
d = sys.argv[1]
agg, images = {}, 0
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
  for f in glob.glob('%s/%s/**/*counter_collection.csv' % (d, c), recursive=True):
    for r in csv.DictReader(open(f)):
      if r['Counter_Name'] != c:
        continue
      n = r['Kernel_Name']
      n = re.sub(r'^void ', '', n); n = re.sub(r'epos::\(anonymous namespace\)::', '', n); n = n.split('(')[0][:58]
      a = agg.setdefault(n, {'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0, 'calls': 0})
      a[c] += float(r['Counter_Value']) * (2 if c == 'FETCH_SIZE' else 1) * 1024
      if c == 'FETCH_SIZE':
        a['calls'] += 1
        if 'im2col3x3' in n:
          images += 1
images = max(images, 1)
tot = sum(a['FETCH_SIZE'] + a['WRITE_SIZE'] for a in agg.values())
print('%d images; %.2f GB per image over all kernels (fetch x2 on gfx950 + write)' % (images, tot / images / 1e9))
for n, a in sorted(agg.items(), key=lambda kv: -(kv[1]['FETCH_SIZE'] + kv[1]['WRITE_SIZE']))[:28]:
  print('%-58s %6.1f launches/img  read %8.1f MB  write %8.1f MB per image  (%5.1f %%)' % (
      n, a['calls'] / images, a['FETCH_SIZE'] / images / 1e6, a['WRITE_SIZE'] / images / 1e6,
      100.0 * (a['FETCH_SIZE'] + a['WRITE_SIZE']) / tot))
PY
rm -rf $D
