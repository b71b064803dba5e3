set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3l
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_t
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_t -- python /root/repo/bench.py --steps 6 --warmup 2 --pipeline-depth 1 --no-cpu-baseline --no-roofline --no-stage-times --traffic off > /root/repo/gpurun_out/r3l/bench.log 2>&1
find /tmp/prof_t -name '*.csv' | head
for f in $(find /tmp/prof_t -name '*kernel_trace.csv'); do python - "$f" <<'P'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
out=open('/root/repo/gpurun_out/r3l/kernel_seq.txt','w')
t0=int(rows[0]['Start_Timestamp'])
for r in rows[-1500:]:
  out.write('%10.1f %8.1f q%s g%s wg%s %s\n'%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r.get('Queue_Id',''),r.get('Grid_Size_X', r.get('Grid_Size','')),r.get('Workgroup_Size_X',r.get('Workgroup_Size','')),r['Kernel_Name'][:70]))
P
done
for f in $(find /tmp/prof_t -name '*memory_copy_trace.csv'); do tail -300 $f > /root/repo/gpurun_out/r3l/memcopy_tail.csv; wc -l $f; done
