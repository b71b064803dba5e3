# same-box A/B of the depthwise workgroup size (EPOS_DW_THREADS): smaller workgroups
# balance 1.67 workgroups per CU better; fewer waves per workgroup share less
for rep in 1 2; do
for t in 256 128 64; do
  EPOS_DW_THREADS=$t python bench.py --no-cpu-baseline --no-roofline --no-stage-times --steps 400 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('dw threads=$t',d['value'],d['ms_per_step'])"
done; done
for t in 256 128 64; do echo "threads $t"; EPOS_DW_THREADS=$t python tools/bench_dw.py 2>/dev/null | tail -8; done
