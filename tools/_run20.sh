cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --traffic static > gpurun_out/r4d/bench_default.json 2>/dev/null
EPOS_HIP_LIB=/root/repo/epos_amd/lib/libepos_hip_dwlight.so python bench.py --steps 60 --warmup 5 --no-cpu-baseline --traffic static > gpurun_out/r4d/bench_dwlight.json 2>/dev/null
python - <<'P'
import json
for n in ('bench_default','bench_dwlight'):
  j=json.loads(open('/root/repo/gpurun_out/r4d/%s.json'%n).read().strip().splitlines()[-1])
  i=j['roofline'].get('in_step') or {}
  print(n, j['value'], j['ms_per_step'], i.get('gemm_ms_per_step'), i.get('depthwise_ms_per_step'), i.get('rest_ms_per_step'))
P
