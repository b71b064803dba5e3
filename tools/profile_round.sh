#!/bin/bash
# Collects the measurements kept under profiles/<round>/ (run on the GPU box through gpurun):
#   bash tools/profile_round.sh r01
# bench JSON lines (default / depth 1 / sparse heads), rocprofv3 kernel-trace stats at
# depth 1 and 3, and two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) for the HBM-side
# traffic of the GEMM kernels. PMC passes never share a run with trace options.
set -u
R=${1:-r01}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
python bench.py --steps 100 --warmup 10 > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log > $OUT/bench_final_default.json
python bench.py --steps 100 --warmup 10 --pipeline-depth 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_final_depth1.json
python bench.py --steps 100 --warmup 10 --sparse-heads --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_final_sparse_heads.json
for d in 1 3; do
  mkdir -p $OUT/kt$d
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$d -- \
    python bench.py --steps 20 --warmup 3 --pipeline-depth $d --no-cpu-baseline > $OUT/kt$d.log 2>&1
  f=$(find $OUT/kt$d -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $OUT/rocprofv3_kernel_stats_depth$d.csv
done
for c in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $OUT/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- \
    python bench.py --steps 4 --warmup 1 --pipeline-depth 1 --no-cpu-baseline --no-roofline > $OUT/pmc_$c.log 2>&1
  f=$(find $OUT/pmc_$c -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_traffic.py "$f" $c > $OUT/pmc_$c.json
done
rm -rf $OUT/kt1 $OUT/kt3 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
ls -la $OUT
