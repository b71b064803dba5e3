#!/bin/bash
# Same-box comparison of bench.py argument sets (interleaved, REPS rounds):
#   bash tools/ab_args.sh "" "--pipeline-depth 2" "--sparse-heads --pipeline-depth 4"
# extra environment / common arguments: COMMON="--steps 40 ..." (default below)
COMMON=${COMMON:---steps 40 --warmup 8 --no-cpu-baseline --traffic off}
for rep in $(seq ${REPS:-2}); do for A in "$@"; do
  python bench.py $COMMON $A 2>/tmp/ab_err.txt | tail -1 | python -c "
import json,sys
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('%-44s FAILED: %s' % (sys.argv[1][-44:], open('/tmp/ab_err.txt').read()[-300:])); sys.exit(0)
r=d.get('roofline') or {}; i=r.get('in_step') or {}
s=d.get('serial_depth1') or {}; t=d.get('timed_regions') or {}
print('%-44s %7.1f img/s (%.1f..%.1f) %6.3f ms | serial %6.1f | in-step gemm %.3f dw %.3f rest %.3f | %s W %s MHz' % (
  sys.argv[1][-44:] or 'default', d['value'], t.get('images_per_sec_min',0), t.get('images_per_sec_max',0), d['ms_per_step'], s.get('images_per_sec',0),
  i.get('gemm_ms_per_step',0), i.get('depthwise_ms_per_step',0), i.get('rest_ms_per_step',0), r.get('power_w'), r.get('core_clock_mhz_under_load')))" "$A"
done; done
