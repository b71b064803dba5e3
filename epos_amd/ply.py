"""PLY object-model reader / writer without bop_toolkit -- what
``inout.load_ply`` [EXTERNAL: bop_toolkit_lib/inout.py] gives the reference's
``ObjectModelStore.load_models`` (datagen.py:68-84): a dict with 'pts' [V,3] float64
(model coordinates, mm) and, when present, 'normals', 'colors' [V,3] and 'faces' [F,3].

PARITY UNPINNED: neither bop_toolkit nor a BOP model file is available here; the reader
follows the public PLY specification (header ``format ascii|binary_little_endian|
binary_big_endian 1.0``, ``element`` / ``property`` declarations, list properties for the
faces) and is checked by round trips through its own writer in all three encodings
(tests/test_ply.py). Model files of the BOP datasets use float x y z [nx ny nz]
[uchar red green blue] [float texture_u texture_v] vertices and
``property list uchar int|uint vertex_indices`` (or ``vertex_index``) faces.
"""
import os
import struct

import numpy as np

_TYPES = {
    'char': ('b', 1), 'int8': ('b', 1), 'uchar': ('B', 1), 'uint8': ('B', 1),
    'short': ('h', 2), 'int16': ('h', 2), 'ushort': ('H', 2), 'uint16': ('H', 2),
    'int': ('i', 4), 'int32': ('i', 4), 'uint': ('I', 4), 'uint32': ('I', 4),
    'float': ('f', 4), 'float32': ('f', 4), 'double': ('d', 8), 'float64': ('d', 8),
}


def _read_header(f):
  if f.readline().strip() != b'ply':
    raise ValueError('not a PLY file')
  fmt, elements = None, []
  while True:
    line = f.readline()
    if not line:
      raise ValueError('unterminated PLY header')
    tok = line.decode('ascii', 'replace').split()
    if not tok or tok[0] in ('comment', 'obj_info'):
      continue
    if tok[0] == 'format':
      fmt = tok[1]
    elif tok[0] == 'element':
      elements.append({'name': tok[1], 'count': int(tok[2]), 'props': []})
    elif tok[0] == 'property':
      if tok[1] == 'list':
        elements[-1]['props'].append(('list', tok[4], tok[2], tok[3]))
      else:
        elements[-1]['props'].append(('scalar', tok[2], tok[1]))
    elif tok[0] == 'end_header':
      break
  if fmt not in ('ascii', 'binary_little_endian', 'binary_big_endian'):
    raise ValueError('unsupported PLY format %r' % fmt)
  return fmt, elements


def load_ply(path):
  """Returns {'pts': f64[V,3], ['normals': f64[V,3]], ['colors': f64[V,3]],
  ['texture_uv': f64[V,2]], ['faces': int64[F,3]]}. Faces with more than three
  vertices are fan-triangulated; other elements are skipped."""
  with open(path, 'rb') as f:
    fmt, elements = _read_header(f)
    data = {}
    end = '<' if fmt == 'binary_little_endian' else '>'
    for el in elements:
      scalars = all(p[0] == 'scalar' for p in el['props'])
      n = el['count']
      if fmt == 'ascii':
        rows = []
        for _ in range(n):
          rows.append(f.readline().split())
        if scalars:
          arr = np.array(rows, dtype=np.float64).reshape(n, len(el['props']))
          data[el['name']] = {p[1]: arr[:, i] for i, p in enumerate(el['props'])}
        else:
          lists = {p[1]: [] for p in el['props'] if p[0] == 'list'}
          for r in rows:
            pos = 0
            for p in el['props']:
              if p[0] == 'list':
                cnt = int(r[pos])
                lists[p[1]].append([int(float(x)) for x in r[pos + 1:pos + 1 + cnt]])
                pos += 1 + cnt
              else:
                pos += 1
          data[el['name']] = lists
      elif scalars:
        dt = np.dtype([(p[1], end + _TYPES[p[2]][0]) for p in el['props']])
        arr = np.frombuffer(f.read(dt.itemsize * n), dtype=dt, count=n)
        data[el['name']] = {p[1]: arr[p[1]].astype(np.float64) for p in el['props']}
      else:
        lists = {p[1]: [] for p in el['props'] if p[0] == 'list'}
        for _ in range(n):
          for p in el['props']:
            if p[0] == 'list':
              cf, cs = _TYPES[p[2]]
              cnt = struct.unpack(end + cf, f.read(cs))[0]
              vf, vs = _TYPES[p[3]]
              lists[p[1]].append(list(struct.unpack(end + vf * cnt, f.read(vs * cnt))))
            else:
              f.read(_TYPES[p[2]][1])
        data[el['name']] = lists
  v = data.get('vertex')
  if v is None or not all(k in v for k in ('x', 'y', 'z')):
    raise ValueError('PLY file %s has no vertex x/y/z' % path)
  model = {'pts': np.stack([v['x'], v['y'], v['z']], axis=1)}
  if all(k in v for k in ('nx', 'ny', 'nz')):
    model['normals'] = np.stack([v['nx'], v['ny'], v['nz']], axis=1)
  if all(k in v for k in ('red', 'green', 'blue')):
    model['colors'] = np.stack([v['red'], v['green'], v['blue']], axis=1)
  if all(k in v for k in ('texture_u', 'texture_v')):
    model['texture_uv'] = np.stack([v['texture_u'], v['texture_v']], axis=1)
  fc = data.get('face')
  if fc:
    key = 'vertex_indices' if 'vertex_indices' in fc else (
        'vertex_index' if 'vertex_index' in fc else None)
    if key is not None:
      tris = []
      for poly in fc[key]:
        for i in range(1, len(poly) - 1):
          tris.append((poly[0], poly[i], poly[i + 1]))
      model['faces'] = np.array(tris, np.int64).reshape(-1, 3)
  return model


def save_ply(path, pts, faces=None, normals=None, colors=None,
             fmt='binary_little_endian'):
  """Minimal writer (round-trip tests, synthetic model stores)."""
  pts = np.asarray(pts, np.float64)
  n = pts.shape[0]
  props = ['float x', 'float y', 'float z']
  cols = [pts.astype(np.float32)]
  if normals is not None:
    props += ['float nx', 'float ny', 'float nz']
    cols.append(np.asarray(normals, np.float32))
  if colors is not None:
    props += ['uchar red', 'uchar green', 'uchar blue']
  faces = None if faces is None else np.asarray(faces, np.int64)
  head = ['ply', 'format %s 1.0' % fmt, 'comment written by epos_amd.ply',
          'element vertex %d' % n] + ['property ' + p for p in props]
  if faces is not None:
    head += ['element face %d' % faces.shape[0],
             'property list uchar int vertex_indices']
  head.append('end_header')
  end = '<' if fmt == 'binary_little_endian' else '>'
  with open(path, 'wb') as f:
    f.write(('\n'.join(head) + '\n').encode('ascii'))
    fl = np.concatenate(cols, axis=1)
    for i in range(n):
      if fmt == 'ascii':
        row = ' '.join(repr(float(x)) for x in fl[i])
        if colors is not None:
          row += ' ' + ' '.join(str(int(c)) for c in colors[i])
        f.write((row + '\n').encode('ascii'))
      else:
        f.write(struct.pack(end + 'f' * fl.shape[1], *fl[i]))
        if colors is not None:
          f.write(struct.pack('BBB', *[int(c) for c in colors[i]]))
    if faces is not None:
      for t in faces:
        if fmt == 'ascii':
          f.write(('3 %d %d %d\n' % tuple(t)).encode('ascii'))
        else:
          f.write(struct.pack(end + 'B', 3) + struct.pack(end + 'iii', *[int(x) for x in t]))


# ------------------------------------------------------------ BOP layout ---
# [EXTERNAL: bop_toolkit_lib/dataset_params.py get_model_params] object ids per dataset
# and the model path template <datasets_path>/<dataset>/models[_<type>]/obj_{id:06d}.ply
BOP_OBJ_IDS = {
    'lm': list(range(1, 16)), 'lmo': [1, 5, 6, 8, 9, 10, 11, 12],
    'tless': list(range(1, 31)), 'tudl': list(range(1, 4)), 'tyol': list(range(1, 22)),
    'ruapc': list(range(1, 15)), 'icmi': list(range(1, 7)), 'icbin': list(range(1, 3)),
    'itodd': list(range(1, 29)), 'hbs': [1, 3, 4, 8, 9, 10, 12, 15, 17, 18, 19, 22, 23,
                                         29, 32, 33],
    'hb': list(range(1, 34)), 'ycbv': list(range(1, 22)), 'hope': list(range(1, 29)),
}


def model_path(datasets_path, dataset, obj_id, model_type=None):
  folder = 'models' if model_type is None else 'models_' + model_type
  return os.path.join(datasets_path, dataset, folder, 'obj_%06d.ply' % obj_id)


def load_models(datasets_path, dataset, model_type=None, obj_ids=None):
  """{obj_id: model dict} for a BOP dataset (datagen.py:68-84)."""
  ids = obj_ids if obj_ids is not None else BOP_OBJ_IDS[dataset]
  return {o: load_ply(model_path(datasets_path, dataset, o, model_type)) for o in ids}
