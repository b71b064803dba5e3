#!/usr/bin/env python
"""Per-hardware-queue view of a rocprofv3 --kernel-trace csv: how many kernels each queue
carried, how long it was busy (union of its kernels' intervals) and idle inside the window,
the distribution of the gaps between consecutive kernels of a queue, and how many queues
were busy on average. Written to find out why infer.py's four pipelines deliver less than
bench.py's (round 6): two pipelines on one queue, or queues that sit idle between steps.

  python tools/trace_queues.py <kernel_trace.csv> [skip_frac]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name'])
            for r in rows)
t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * skip
ev = [e for e in ev if e[0] >= t_lo]
t0, t1 = ev[0][0], max(e[1] for e in ev)
wall = t1 - t0


def union(iv):
  tot, cs, ce = 0, None, None
  for s, e in sorted(iv):
    if ce is None or s > ce:
      if ce is not None:
        tot += ce - cs
      cs, ce = s, e
    else:
      ce = max(ce, e)
  if ce is not None:
    tot += ce - cs
  return tot


print('window %.1f ms, %d kernels, %d queues' % (wall / 1e6, len(ev), len(set(e[2] for e in ev))))
busy_sum = 0
for q in sorted(set(e[2] for e in ev)):
  iv = sorted((s, e) for s, e, qq, _ in ev if qq == q)
  b = union(iv)
  busy_sum += b
  gaps = [iv[i + 1][0] - max(x[1] for x in iv[:i + 1][-4:]) for i in range(len(iv) - 1)]
  gaps = [g for g in gaps if g > 0]
  big = [g for g in gaps if g > 100000]                 # > 100 us: between steps
  nets = sum(1 for _, _, qq, n in ev if qq == q and 'im2col3x3' in n)
  print('  queue %-3s %6d kernels, %4d steps (im2col launches); busy %5.1f %% of the window; gaps > 100 us: '
        '%4d, mean %.2f ms, sum %.1f %% of the window' % (
            q, len(iv), nets, 100.0 * b / wall, len(big),
            (sum(big) / len(big) / 1e6) if big else 0.0, 100.0 * sum(big) / wall))
print('queues busy on average: %.2f; any queue busy %.1f %% of the window' % (
    busy_sum / wall, 100.0 * union([(s, e) for s, e, _, _ in ev]) / wall))
