#!/bin/bash
# SQ / LDS counters of ONE shape of the fp16-pair GEMM (separate --pmc passes, no trace options):
#   bash tools/pmc_h2_gemm.sh 16384 1024 4096 [presplit] [tall] > gpurun_out/pmc_h2.json
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=$(echo "$@" | tr ' ' '_')
OUT=gpurun_out/pmc_h2_$TAG; rm -rf $OUT; mkdir -p $OUT
M=$1; N=$2; K=$3; shift 3
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD"; do
  i=$((i+1)); mkdir -p $OUT/p$i
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python tools/bench_one_gemm_h2.py $M $N $K ${PMC_ITERS:-6} "$@" > $OUT/p$i.log 2>&1
done
python - $OUT $M $N $K "$@" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
vals = {}
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    if 'pointwise_gemm_h2' in r['Kernel_Name']:
      a = vals.setdefault(r['Counter_Name'], [0, 0.0, r['Kernel_Name'][:90]]); a[0] += 1; a[1] += float(r['Counter_Value'])
c = {k: v[1] / v[0] for k, v in vals.items()}
kern = next(iter(vals.values()))[2] if vals else ''
d = {}
if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
  d['mfma_busy_cycles_per_simd'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024
  d['gui_active_cycles_per_xcd'] = c['GRBM_GUI_ACTIVE'] / 8
  d['mfma_pipe_utilisation'] = d['mfma_busy_cycles_per_simd'] / d['gui_active_cycles_per_xcd']
if 'SQ_WAVE_CYCLES' in c:
  for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_MISC'):
    if k in c: d[k + '/WAVE_CYCLES'] = c[k] / c['SQ_WAVE_CYCLES']
print(json.dumps({'kernel': kern, 'problem': ' '.join(sys.argv[2:]), 'counters': c, 'derived': d}, indent=1))
PY
