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
# the window: from the step at `skip` of all steps (a step = one im2col3x3 launch, the
# network's first kernel) to the step at 95 % -- set-up, warm-up and the drain stay outside
steps = [e[0] for e in ev if 'im2col3x3' in e[3]]
if len(steps) >= 20:
  t_lo, t_hi = steps[int(len(steps) * skip)], steps[int(len(steps) * 0.95)]
  n_steps = int(len(steps) * 0.95) - int(len(steps) * skip)
else:
  t_lo, t_hi, n_steps = ev[0][0] + (ev[-1][1] - ev[0][0]) * skip, ev[-1][1], 0
ev = [e for e in ev if t_lo <= e[0] < t_hi]
t0, t1 = t_lo, t_hi
wall = t1 - t0
if n_steps:
  print('%d steps in the window: %.3f ms per step' % (n_steps, wall / n_steps / 1e6))


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
  big = [g for g in gaps if g > 50000]                  # > 50 us: between steps
  nets = sum(1 for _, _, qq, n in ev if qq == q and 'im2col3x3' in n)
  print('  queue %-3s %6d kernels, %4d steps (im2col launches); busy %5.1f %% of the window; gaps > 50 us: '
        '%4d, mean %.2f ms, sum %.1f %% of the window' % (
            q, len(iv), nets, 100.0 * b / wall, len(big),
            (sum(big) / len(big) / 1e6) if big else 0.0, 100.0 * sum(big) / wall))
print('queues busy on average: %.2f; any queue busy %.1f %% of the window' % (
    busy_sum / wall, 100.0 * union([(s, e) for s, e, _, _ in ev]) / wall))
