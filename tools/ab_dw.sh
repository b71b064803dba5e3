# same-box A/B of depthwise-kernel shapes (register footprint vs loads per output):
#   python -c "from epos_amd import build; build.build_variant('dwL1', ['-DEPOS_DW_L=1','-DEPOS_DW_MIN_BLOCKS=5']); build.build_variant('dwL2', ['-DEPOS_DW_L=2','-DEPOS_DW_MIN_BLOCKS2=4'])"
for rep in 1 2; do
for v in "" "dwL1:1" "dwL2:2" ":1"; do
  lib=${v%%:*}; rows=${v##*:}
  if [ -n "$lib" ]; then export EPOS_HIP_LIB=/root/repo/epos_amd/lib/libepos_hip_$lib.so; else unset EPOS_HIP_LIB; fi
  if [ -n "$rows" ] && [ "$v" != "" ]; then export EPOS_DW_ROWS=$rows; else unset EPOS_DW_ROWS; fi
  python bench.py --no-cpu-baseline --no-roofline --no-stage-times --steps 400 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('variant=[$v]',d['value'],d['ms_per_step'])"
done; done
