for rep in 1 2; do
for v in "" gc256 gc512; do
  if [ -n "$v" ]; then export EPOS_HIP_LIB=/root/repo/epos_amd/lib/libepos_hip_$v.so; else unset EPOS_HIP_LIB; fi
  python bench.py --no-cpu-baseline --no-roofline --steps 600 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('variant=[$v]',d['value'],d['ms_per_step'],d['serial_depth1']['images_per_sec'],d['serial_depth1']['stage_ms']['fitting'])"
done; done
