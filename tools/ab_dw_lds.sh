# same-box A/B: unused LDS reserved per depthwise workgroup (occupancy throttle), so that a CU
# holds "one GEMM workgroup + one depthwise workgroup" instead of three depthwise ones
for rep in 1 2; do
for kb in 0 32 48 64; do
  EPOS_DW_LDS_KB=$kb python bench.py --no-cpu-baseline --no-roofline --no-stage-times --steps 400 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('dw lds pad ${kb} KB',d['value'],d['ms_per_step'])"
done; done
