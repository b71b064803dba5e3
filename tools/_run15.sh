set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3p
export TMPDIR=/tmp

cd /tmp
for v in gs16_1 gs16_1_noload gs16_1_nomath gs16_2 gs16_2_noload; do
rm -rf /tmp/prof_a
if [ $v = default ]; then unset EPOS_HIP_LIB; else export EPOS_HIP_LIB=/root/repo/epos_amd/lib/libepos_hip_$v.so; fi
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -- python /root/repo/bench.py --steps 30 --warmup 5 --pipeline-depth 1 --no-cpu-baseline --no-roofline --no-stage-times --traffic off > /dev/null 2>&1
f=$(find /tmp/prof_a -name '*kernel_stats.csv' | head -1)
echo "== $v"
python - $f <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
  if 'ransac_gc' in r['Name']:
    print(r['Name'][:60], r['Calls'], '%.1f'%(float(r['AverageNs'])/1e3), r['MinNs'], r['MaxNs'])
P
done
