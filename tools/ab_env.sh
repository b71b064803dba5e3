#!/bin/bash
# Same-box A/B of two environments of bench.py: bash tools/ab_env.sh "VAR=a" "VAR=b" [bench args]
# (interleaved runs A B A B)
A=$1; B=$2; shift 2
for rep in 1 2; do for E in "$A" "$B"; do
  env $E python bench.py --steps 60 --warmup 8 --no-cpu-baseline --traffic static "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d['roofline'].get('in_step') or {}
s=d.get('serial_depth1') or {}
print('%-36s %7.1f img/s %6.3f ms | serial %6.1f (pred %.3f) | in-step gemm %.3f dw %.3f rest %.3f | per-launch %.1f us' % (
  sys.argv[1][-36:], d['value'], d['ms_per_step'], s.get('images_per_sec',0), (s.get('stage_ms') or {}).get('prediction',0),
  i.get('gemm_ms_per_step',0), i.get('depthwise_ms_per_step',0), i.get('rest_ms_per_step',0), d['roofline'].get('avg_launch_us',0)))" "$E"
done; done
