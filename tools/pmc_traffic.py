#!/usr/bin/env python
"""Per-kernel-family average of one rocprofv3 --pmc counter (counter_collection.csv):
  python tools/pmc_traffic.py <csv> <COUNTER>      -> JSON {family: {launches, avg, sum}}
  python tools/pmc_traffic.py --merge fetch.json write.json
      -> the gemm_hbm_traffic_pmc.json bench.py reports as roofline.traffic: HBM-side bytes per
         GEMM launch = 2 x FETCH_SIZE (gfx950: the counter tallies 128-B requests at 64 B,
         MI355X_MICROARCH.md, HBM) + WRITE_SIZE, KB -> bytes."""
import csv, json, sys
if sys.argv[1] == '--merge':
  f = json.load(open(sys.argv[2]))['groups']; w = json.load(open(sys.argv[3]))['groups']
  g, d = 'pointwise_gemm', 'depthwise'
  out = {
      'kernel': 'pointwise_gemm_h2_f32 / pointwise_gemm_split_f32 / pointwise_gemm_dma_f32 / pointwise_gemv_f32 (all GEMM launches of the plan)',
      'launches': f[g]['launches'], 'fetch_size_kb_avg': f[g]['avg'], 'write_size_kb_avg': w[g]['avg'],
      'correction': 'gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> doubled; '
                    'WRITE_SIZE uncorrected; KB -> bytes x1024',
      'traffic_bytes_per_launch': (2 * f[g]['avg'] + w[g]['avg']) * 1024,
      'note': 'two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over `bench.py --steps 4 --warmup 1 '
              '--pipeline-depth 1` (tools/profile_round.sh); the counters sit on the fabric side of L2 and include '
              'Infinity-Cache hits',
  }
  total = sum((2 * f[k]['sum'] + w.get(k, {'sum': 0})['sum']) for k in f) * 1024
  imgs = 5            # steps 4 + warm-up 1 (+ the set-up pass of the one plan is excluded by launches / 72)
  out['gemm_launches_per_image'] = 72
  out['traffic_mb_per_image_all_kernels'] = total / (f[g]['launches'] / 72.0) / 1e6
  if d in f:
    out['depthwise_fetch_kb_avg'] = f[d]['avg']; out['depthwise_write_kb_avg'] = w[d]['avg']
    out['depthwise_traffic_bytes_per_launch'] = (2 * f[d]['avg'] + w[d]['avg']) * 1024
  print(json.dumps(out, indent=1))
  sys.exit(0)
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
