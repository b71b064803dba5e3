#!/usr/bin/env python
"""Timeline statistics of a rocprofv3 --kernel-trace csv: GPU busy fraction (union of
kernel intervals), average concurrency, time by kernel family, and how much of the GEMM
time overlaps another GEMM.  python tools/trace_overlap.py <kernel_trace.csv> [skip_frac]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * skip          # drop the warm-up / calibration part
ev = [e for e in ev if e[0] >= t_lo]
t0, t1 = ev[0][0], max(e[1] for e in ev)
def fam(n):
  return ('gemm' if 'pointwise_gem' in n else 'dw' if 'depthwise' in n else
          'ransac' if 'ransac' in n else 'corr' if 'corr_' in n else 'other')
def union(iv):
  tot, cur_s, cur_e = 0, None, None
  for s, e in sorted(iv):
    if cur_e is None or s > cur_e:
      if cur_e is not None: tot += cur_e - cur_s
      cur_s, cur_e = s, e
    else:
      cur_e = max(cur_e, e)
  if cur_e is not None: tot += cur_e - cur_s
  return tot
wall = t1 - t0
busy = union([(s, e) for s, e, _ in ev])
tot = sum(e - s for s, e, _ in ev)
print('window %.1f ms, %d kernels; busy %.1f %%, sum of durations / wall = %.2f' % (
    wall / 1e6, len(ev), 100.0 * busy / wall, tot / wall))
for f in ['gemm', 'dw', 'ransac', 'corr', 'other']:
  iv = [(s, e) for s, e, n in ev if fam(n) == f]
  if iv:
    print('  %-7s n=%5d  sum %.1f ms (%.1f %% of wall)  union %.1f %% of wall  avg %.1f us' % (
        f, len(iv), sum(e - s for s, e in iv) / 1e6, 100.0 * sum(e - s for s, e in iv) / wall,
        100.0 * union(iv) / wall, sum(e - s for s, e in iv) / len(iv) / 1e3))
