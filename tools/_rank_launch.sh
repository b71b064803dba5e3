#!/usr/bin/env bash
# torchrun --no-python helper of tools/run_scale.sh: pins this rank to its share of the host
# cores, then execs the command. Usage: _rank_launch.sh <ncpu> <nranks> <cmd...>
ncpu=$1; n=$2; shift 2
r=${LOCAL_RANK:-0}
per=$((ncpu / n)); [ "$per" -ge 1 ] || per=1
lo=$((r * per)); hi=$((lo + per - 1))
if command -v taskset >/dev/null 2>&1; then exec taskset -c "$lo-$hi" "$@"; fi
exec "$@"
