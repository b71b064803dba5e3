#!/usr/bin/env python
"""Per-kernel average of one rocprofv3 --pmc counter (counter_collection.csv):
python tools/pmc_traffic.py <csv> <COUNTER>  ->  JSON {kernel group: {launches, avg}}"""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
name = sys.argv[2]
groups = {}
for r in rows:
  if r.get('Counter_Name') != name:
    continue
  k = r['Kernel_Name']
  g = ('pointwise_gemm' if 'pointwise_gemm' in k or 'pointwise_gemv' in k else
       'depthwise' if 'depthwise' in k else 'other')
  a = groups.setdefault(g, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
print(json.dumps({'counter': name, 'groups': {g: {'launches': n, 'avg': s / n, 'sum': s}
                                              for g, (n, s) in groups.items()}}))
