#!/bin/bash
# Where infer.py's pipelined loop loses against bench.py (round 6): the same dataset through
# infer.py with one thing changed at a time, and the hardware-queue view of both loops.
#   bash tools/infer_diag.sh <out dir>      (GPU box; writes <out dir>/infer_diag.txt)
OUT=${1:-gpurun_out/infer_diag}
mkdir -p $OUT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
D=/tmp/epos_e2e_diag
python tools/infer_end_to_end.py --frames 600 --keep $D --no-bench > $OUT/e2e_keep.log 2>&1
export TF_MODELS_PATH=$D/models TF_DATA_PATH=$D/data
run() {   # tag, env..., -- args
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python infer.py --model=ycbv-xc65 --infer_tfrecord_names=ycbv_test --infer_name $tag --sparse_heads false "$@" > $OUT/$tag.log 2>&1
  echo "$tag: $(grep '^Throughput' $OUT/$tag.log | sed 's/(inference loop.*//') | $(grep '^Host time' $OUT/$tag.log)"
}
{
run base X=1 --
run upload_copy EPOS_UPLOAD=copy --
run no_events EPOS_INFER_TIMING=0 --
run no_events_copy EPOS_INFER_TIMING=0 EPOS_UPLOAD=copy --
run depth6 X=1 -- --pipeline_depth 6
run depth5 X=1 -- --pipeline_depth 5
run base_again X=1 --
} | tee $OUT/infer_diag.txt
mkdir -p $OUT/kt_infer $OUT/kt_bench
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_infer -- python infer.py --model=ycbv-xc65 --infer_tfrecord_names=ycbv_test --infer_name kt --sparse_heads false > $OUT/kt_infer.log 2>&1
f=$(find $OUT/kt_infer -name '*kernel_trace.csv' | head -1)
{ echo "== infer.py under rocprofv3: $(grep '^Throughput' $OUT/kt_infer.log | sed 's/(inference loop.*//')"; [ -n "$f" ] && python tools/trace_queues.py "$f" 0.5; } | tee -a $OUT/infer_diag.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_bench -- python bench.py --steps 100 --warmup 10 --timed-repeats 2 --no-cpu-baseline --traffic static --no-stage-times --no-roofline > $OUT/kt_bench.log 2>&1
f=$(find $OUT/kt_bench -name '*kernel_trace.csv' | head -1)
{ echo "== bench.py under rocprofv3: $(tail -1 $OUT/kt_bench.log | cut -c1-80)"; [ -n "$f" ] && python tools/trace_queues.py "$f" 0.5; } | tee -a $OUT/infer_diag.txt
rm -rf $OUT/kt_infer $OUT/kt_bench
