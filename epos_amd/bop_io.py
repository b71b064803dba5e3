"""BOP'19 result files -- restates what ``bop_toolkit_lib.inout.save_bop_results(
path, poses, version='bop19')`` writes (called at scripts/infer.py:753-760; the
bop_toolkit submodule is empty, format per the BOP'19 spec cited at
infer.py:751-752): a header line and one row per estimate,
``scene_id,im_id,obj_id,score,R,t,time`` with R (9 values, row-major) and t (3
values, mm) space separated."""
import numpy as np


def save_bop_results(path, results, version='bop19'):
  if version != 'bop19':
    raise ValueError('Unknown version of BOP results.')
  lines = ['scene_id,im_id,obj_id,score,R,t,time']
  for res in results:
    run_time = res['time'] if 'time' in res else -1
    lines.append('{scene_id},{im_id},{obj_id},{score},{R},{t},{time}'.format(
        scene_id=res['scene_id'], im_id=res['im_id'], obj_id=res['obj_id'],
        score=res['score'],
        R=' '.join(map(str, np.asarray(res['R']).flatten().tolist())),
        t=' '.join(map(str, np.asarray(res['t']).flatten().tolist())),
        time=run_time))
  with open(path, 'w') as f:
    f.write('\n'.join(lines))


def load_bop_results(path, version='bop19'):
  if version != 'bop19':
    raise ValueError('Unknown version of BOP results.')
  results = []
  with open(path, 'r') as f:
    for i, line in enumerate(f):
      if i == 0 or not line.strip():
        continue
      e = line.strip().split(',')
      results.append({
          'scene_id': int(e[0]), 'im_id': int(e[1]), 'obj_id': int(e[2]),
          'score': float(e[3]),
          'R': np.array(list(map(float, e[4].split())), np.float64).reshape(3, 3),
          't': np.array(list(map(float, e[5].split())), np.float64).reshape(3, 1),
          'time': float(e[6])})
  return results
