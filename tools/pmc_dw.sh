#!/bin/bash
# SQ / TCP counters of the depthwise kernel on the network's shapes (tools/bench_dw.py --h2;
# separate --pmc passes, no trace options):   bash tools/pmc_dw.sh > gpurun_out/pmc_dw.json
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_dw; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" \
           "SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES"; do
  i=$((i+1)); mkdir -p $OUT/p$i
  PMC_DW=1 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python tools/bench_dw.py --h2 --one > $OUT/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, json
vals = {}
for f in glob.glob('gpurun_out/pmc_dw/p*/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    if 'depthwise3x3_s1' in r['Kernel_Name']:
      a = vals.setdefault(r['Counter_Name'], [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
c = {k: v[1] / v[0] for k, v in vals.items()}
d = {}
if 'SQ_WAVE_CYCLES' in c and 'SQ_WAVES' in c and c['SQ_WAVES']:
  d['cycles_per_wave'] = c['SQ_WAVE_CYCLES'] / c['SQ_WAVES'] * 4   # SQ_WAVE_CYCLES counts in quad-cycles
if 'SQ_WAIT_INST_ANY' in c and 'SQ_WAVE_CYCLES' in c and c['SQ_WAVE_CYCLES']:
  d['wait_inst_any_frac_of_wave_cycles'] = c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']
if 'SQ_ACTIVE_INST_ANY' in c and 'SQ_WAVE_CYCLES' in c and c['SQ_WAVE_CYCLES']:
  d['active_inst_any_frac_of_wave_cycles'] = c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES']
if 'TCC_HIT_sum' in c and 'TCC_MISS_sum' in c and c['TCC_HIT_sum'] + c['TCC_MISS_sum']:
  d['l2_hit_rate'] = c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum'])
if 'SQ_INSTS_VMEM_RD' in c and 'SQ_WAVES' in c and c['SQ_WAVES']:
  d['vmem_reads_per_wave'] = c['SQ_INSTS_VMEM_RD'] / c['SQ_WAVES']
  d['vmem_writes_per_wave'] = c.get('SQ_INSTS_VMEM_WR', 0) / c['SQ_WAVES']
  d['valu_per_wave'] = c.get('SQ_INSTS_VALU', 0) / c['SQ_WAVES']
print(json.dumps({'kernel': 'depthwise3x3_s1_kernel<4, relu_in, -, 2 rows>, 60 x 80 x 728 rate 2, fp16-pair output; per dispatch averages',
                  'counters': c, 'derived': d}, indent=1))
PY
