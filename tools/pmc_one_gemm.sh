#!/bin/bash
# SQ / LDS counters of ONE GEMM shape (separate --pmc passes, no trace options):
#   bash tools/pmc_one_gemm.sh 4800 728 728 > gpurun_out/pmc_gemm.json
#   BENCH_SPLIT=1 bash tools/pmc_one_gemm.sh 19200 728 728   (split-operand kernel)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_one; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1)); mkdir -p $OUT/p$i
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python tools/bench_one_gemm.py $1 $2 $3 ${PMC_ITERS:-5} > $OUT/p$i.log 2>&1
done
python - "$@" <<'PY'
import csv, glob, json, sys
vals = {}
for f in glob.glob('gpurun_out/pmc_one/p*/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    if 'pointwise_gemm' in r['Kernel_Name']:
      a = vals.setdefault(r['Counter_Name'], [0, 0.0, r['Kernel_Name'][:80]]); a[0] += 1; a[1] += float(r['Counter_Value'])
c = {k: v[1] / v[0] for k, v in vals.items()}
kern = next(iter(vals.values()))[2] if vals else ''
d = {}
if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
  d['mfma_busy_cycles_per_simd'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024
  d['gui_active_cycles_per_xcd'] = c['GRBM_GUI_ACTIVE'] / 8
  d['mfma_pipe_utilisation'] = d['mfma_busy_cycles_per_simd'] / d['gui_active_cycles_per_xcd']
print(json.dumps({'kernel': kern, 'problem': 'M=%s N=%s K=%s, per dispatch averages over 5 dispatches (cold clocks)' % tuple(sys.argv[1:4]),
                  'counters': c, 'derived': d}, indent=1))
PY
