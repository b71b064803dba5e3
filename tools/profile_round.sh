#!/bin/bash
# Collects the measurements kept under profiles/<round>/ (run on the GPU box through gpurun):
#   bash tools/profile_round.sh r02
# - the bench line at the DRIVER'S command (bench.py --gpus 1 --steps 20 --warmup 5) and at
#   100 steps, depth 1, sparse heads, fp32-MFMA GEMMs;
# - rocprofv3 --kernel-trace --stats of the driver's command (pipeline depth 4) and of
#   depth 1, plus the per-kernel-family union of busy time per step from the depth-4 trace
#   (so that sum(kernel time per step) <= ms_per_step can be checked, not argued);
# - two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) merged into gemm_hbm_traffic_pmc.json
#   (gfx950 x2 correction on FETCH_SIZE; PMC passes never share a run with trace options);
# - the RANSAC-only microbenchmark.
set -u
R=${1:-r05}
cd "$(dirname "$0")/.."
OUT=gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/bench_driver_cmd.err | tail -1 > $OUT/bench_driver_cmd.json
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --traffic static 2>/dev/null | tail -1 > $OUT/bench_driver_cmd_again.json
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --traffic static 2>/dev/null | tail -1 > $OUT/bench_100steps.json
python bench.py --steps 100 --warmup 10 --sparse-heads --no-cpu-baseline --traffic static 2>/dev/null | tail -1 > $OUT/bench_sparse_heads.json
EPOS_HIP_LIB=$PWD/epos_amd/lib/libepos_hip_ref.so EPOS_GEMM_SPLIT=0 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --traffic static 2>/dev/null | tail -1 > $OUT/bench_fp32_mfma.json
EPOS_GEMM_H2=0 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --traffic static 2>/dev/null | tail -1 > $OUT/bench_bf16x6.json
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --traffic static 2>/dev/null | tail -1 > $OUT/bench_100steps_b.json
# the other BASELINE configurations
python bench.py --steps 60 --warmup 5 --num-objs 1 --objs-per-image 1 --no-cpu-baseline --traffic static 2>/dev/null | tail -1 > $OUT/bench_c1.json
python bench.py --steps 40 --warmup 5 --height 540 --width 720 --num-objs 30 --objs-per-image 8 --instances 2 --no-cpu-baseline --traffic static 2>/dev/null | tail -1 > $OUT/bench_c4.json
python bench.py --steps 20 --warmup 3 --model-variant resnet_v1_101_beta --num-objs 15 --batch-per-gpu 8 --pipeline-depth 2 --no-cpu-baseline --traffic static 2>/dev/null | tail -1 > $OUT/bench_c5.json
python bench.py --steps 40 --warmup 5 --batch-per-gpu 4 --no-cpu-baseline --traffic static 2>/dev/null | tail -1 > $OUT/bench_c3_shard_batch4.json
for d in 4 1; do
  mkdir -p $OUT/kt$d
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$d -- \
    python bench.py --gpus 1 --steps 20 --warmup 5 --timed-repeats 1 --pipeline-depth $d --no-cpu-baseline --traffic static --no-stage-times > $OUT/kt$d.log 2>&1
  f=$(find $OUT/kt$d -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $OUT/rocprofv3_kernel_stats_depth$d.csv
  f=$(find $OUT/kt$d -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python tools/trace_overlap.py "$f" 0.5 > $OUT/kernel_trace_busy_union_depth$d.txt
  grep "^{\"metric\"" $OUT/kt$d.log | tail -1 > $OUT/bench_under_rocprof_depth$d.json
done
for c in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $OUT/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- \
    python bench.py --steps 4 --warmup 1 --timed-repeats 1 --pipeline-depth 1 --no-cpu-baseline --no-roofline --no-stage-times > $OUT/pmc_$c.log 2>&1
  f=$(find $OUT/pmc_$c -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_traffic.py "$f" $c > $OUT/pmc_$c.json
done
python tools/pmc_traffic.py --merge $OUT/pmc_FETCH_SIZE.json $OUT/pmc_WRITE_SIZE.json > $OUT/gemm_hbm_traffic_pmc.json
python tools/bench_ransac.py > $OUT/ransac_microbench.txt 2>&1
# round 6: the planted workload (fitting stage with EPOS-like inlier ratios) next to the default
for f in 0.3 0.5 0.7; do
  python bench.py --planted-poses --planted-outliers $f --steps 60 --warmup 8 --no-cpu-baseline --traffic static 2>/dev/null | tail -1 > $OUT/bench_planted_c2_$f.json
done
python bench.py --planted-poses --planted-outliers 0.5 --height 540 --width 720 --num-objs 30 --objs-per-image 8 --instances 2 --steps 40 --warmup 5 --no-cpu-baseline --traffic static 2>/dev/null | tail -1 > $OUT/bench_planted_c4_0.5.json
python bench.py --weights heavy-tailed --steps 60 --warmup 8 --no-cpu-baseline --traffic static 2>/dev/null | tail -1 > $OUT/bench_weights_heavy_tailed.json
rm -rf $OUT/kt1 $OUT/kt4 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
ls -la $OUT
