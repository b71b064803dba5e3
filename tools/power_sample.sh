#!/bin/bash
# Samples socket power / clocks (rocm-smi) while a command runs:
#   bash tools/power_sample.sh gpurun_out/power.txt python bench.py --steps 2000 --no-cpu-baseline
OUT=$1; shift
"$@" > $OUT.cmd.log 2>&1 &
PID=$!
sleep 25       # import torch + plan build + calibration
for i in $(seq 1 12); do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' ' >> $OUT
  echo >> $OUT
  sleep 0.5
done
wait $PID
tail -c 600 $OUT.cmd.log | head -c 600 >> $OUT
