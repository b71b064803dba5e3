#!/bin/bash
# L1 (TCP) -> L2 (TCC) read requests of ONE fp16-pair GEMM launch, next to a plain copy of a
# tensor of known size for the bytes-per-request calibration (round 6: is the A operand's
# 64-byte-per-row K step fetched from L2 twice per 128-byte line?).
#   bash tools/pmc_gemm_l1.sh M N K > gpurun_out/pmc_gemm_l1.txt
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
M=${1:-4800}; N=${2:-728}; K=${3:-728}
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_READ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_WRITE_REQ_sum TCC_EA0_RDREQ_sum"; do
  d=$(mktemp -d /tmp/pmcl1_XXXX)
  rocprofv3 --pmc $set --output-format csv -d $d -- python tools/bench_one_gemm_h2.py $M $N $K 5 presplit --copy > $d/log.txt 2>&1
  python - "$d" "$M" "$N" "$K" <<'PY'
import csv, glob, sys
d, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
acc = {}
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    kind = 'gemm' if 'pointwise_gemm_h2' in r['Kernel_Name'] else 'copy' if ('copy' in r['Kernel_Name'].lower() or 'elementwise' in r['Kernel_Name'].lower()) else None
    if kind:
      a = acc.setdefault((kind, r['Counter_Name']), [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
for (kind, c), (n, v) in sorted(acc.items()):
  print('%-5s %-30s %14.0f per launch (%d launches)' % (kind, c, v / n, n))
if not acc:
  print(open(d + '/log.txt').read()[-400:])
PY
  rm -rf $d
done
echo "GEMM $M x $N x $K, A pre-split: the lanes request A $(python -c "print(round($M*$K*4*(( $N+127)//128)/1e6,1))") MB (x column tiles) + W $(python -c "print(round($K*(( $N+127)//128)*128*4*(($M+127)//128)/1e6,1))") MB (x row tiles); the copy moves 2 x $(python -c "print(round($M*$K*4/1e6,1))") MB"
