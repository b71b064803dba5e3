#!/bin/bash
# Fabric-side bytes of ONE depthwise launch on the middle-flow tensor (60 x 80 x 728, rate 2,
# fp16-pair output), rows of 728 floats vs rows padded to 736 (= 23 lines of 128 bytes):
# FETCH_SIZE and WRITE_SIZE in separate passes (gfx950: FETCH_SIZE x 2, MI355X_MICROARCH.md).
#   bash tools/pmc_dw_traffic.sh > gpurun_out/pmc_dw_traffic.txt
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for pad in "" "--pad32"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$(mktemp -d /tmp/pmcdw_XXXX)
    rocprofv3 --pmc $c --output-format csv -d $d -- python tools/bench_dw.py --h2 --one $pad > $d/log.txt 2>&1
    python - "$d" "$c" "$pad" <<'PY'
import csv, glob, sys
d, c, pad = sys.argv[1:4]
v = [float(r['Counter_Value']) for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True)
     for r in csv.DictReader(open(f)) if 'depthwise3x3_s1' in r['Kernel_Name'] and r['Counter_Name'] == c]
if v:
  kb = sum(v) / len(v) * (2 if c == 'FETCH_SIZE' else 1)
  print('rows %s: %s = %.2f MB per launch (%d launches; KB counter%s); algorithmic 13.98 MB' % (
      'padded to 736 floats' if pad else 'of 728 floats', c, kb / 1024, len(v), ', x2 on gfx950' if c == 'FETCH_SIZE' else ''))
else:
  print('no samples', c, pad, open(d + '/log.txt').read()[-300:])
PY
    rm -rf $d
  done
done
