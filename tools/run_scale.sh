#!/usr/bin/env bash
# 1 -> N GPU scaling run of bench.py on ONE node (BASELINE config C3: images sharded over the
# GPUs, 4 images per GPU and step, ONE RCCL all_gather of pose records per timed region).
#
#     tools/run_scale.sh [N ...]          # default: 1 2 4 8 (those <= the visible devices)
#
# One rank per device (torchrun, rendezvous on 127.0.0.1), each rank pinned to the CPU cores
# of its share of the socket(s) (the host side of a step is ~0.3 ms of Python + one pinned
# D2H copy per step: ranks that migrate between NUMA nodes lose 5-10 %), eight HIP hardware
# queues per process (one per pipeline stream, bench.py sets it too), dmabuf IPC for RCCL.
# Checks per N: the line's n_gpus, config.dist_backend == nccl, config.rccl_ranks_seen == N;
# prints the aggregate images/s, the per-rank figure and the efficiency vs N = 1.
set -euo pipefail
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=8
ndev=$(python -c 'import torch; print(torch.cuda.device_count())')
[ "$ndev" -ge 1 ] || { echo "no HIP device"; exit 1; }
ns=("$@"); [ ${#ns[@]} -gt 0 ] || ns=(1 2 4 8)
steps=${STEPS:-40}; warm=${WARMUP:-5}
ncpu=$(nproc)
out=${OUT:-gpurun_out/scale}; mkdir -p "$out"
base=""
for n in "${ns[@]}"; do
  [ "$n" -le "$ndev" ] || { echo "N=$n: only $ndev device(s) visible, skipped"; continue; }
  port=$((29500 + n))
  log="$out/bench_n$n.json"
  if [ "$n" -eq 1 ]; then
    python bench.py --gpus 1 --steps "$steps" --warmup "$warm" --batch-per-gpu 4 \
        --no-cpu-baseline --traffic off > "$log"
  else
    # per-rank affinity: rank r gets cores [r*ncpu/n, (r+1)*ncpu/n) (tools/_rank_launch.sh)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" \
        --master-addr 127.0.0.1 --master-port "$port" --no-python \
        tools/_rank_launch.sh "$ncpu" "$n" python bench.py --gpus "$n" --steps "$steps" \
        --warmup "$warm" --no-cpu-baseline --traffic off > "$log"
  fi
  python - "$log" "$n" "$base" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith('{"metric"')]
assert len(line) == 1, 'rank 0 must print exactly one line'
d = json.loads(line[0]); n = int(sys.argv[2])
assert d['n_gpus'] == n, d['n_gpus']
if n > 1:
  assert d['config']['dist_backend'] == 'nccl', d['config']['dist_backend']
  assert d['config']['rccl_ranks_seen'] == n, d['config']['rccl_ranks_seen']
base = float(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] else d['value']
print('N=%d  %.1f images/s aggregate  %.1f per rank  efficiency vs N=1: %.3f  (%.3f ms/step, '
      'batch %d per GPU)' % (n, d['value'], d['value'] / n, d['value'] / n / base,
                             d['ms_per_step'], d['config']['batch_per_gpu']))
PY
  if [ "$n" -eq 1 ]; then base=$(python -c "import json,sys; print(json.loads([l for l in open('$log') if l.startswith('{\"metric\"')][0])['value'])"); fi
done
