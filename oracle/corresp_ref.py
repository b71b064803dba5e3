"""ORACLE (test infrastructure, not product): numpy restatement of
``epos_lib/corresp.py::establish_many_to_many`` (corresp.py:9-101) and
``epos_lib/misc.py::convert_px_indices_to_im_coords`` (misc.py:14-26).

PINNED: checked bit-for-bit against golden vectors produced by the *imported*
reference function (tests/golden/make_golden.py, run in the build container where
/root/reference exists; fixtures committed under tests/golden/).

Written as explicit per-pixel loops over the arithmetic (not as a re-use of the
reference's fancy-indexing expressions) so that it states dtype and rounding of
every step the HIP kernel must reproduce:

  mask      : obj_confs[y, x, obj_id] > float32(min_obj_conf)          (:46-47)
  order     : raster over masked pixels, then ascending fragment id     (:52,:67)
  coord_2d  : float64 (x + 0.5, y + 0.5) * (1 / output_scale)           (:55-57)
  frag keep : conf > float32(max_conf * float32(min_frag_rel_conf))     (:63-64)
  coord_3d  : center_f64 + float64(float32(float64(loc_f32) * size_f64)) (:71-78)
  conf      : float32(conf_obj * conf_frag)                             (:82-84)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
import numpy as np


def establish_many_to_many(obj_confs, frag_confs, frag_coords, gt_obj_ids,
                           obj_ids, frag_centers, frag_sizes, output_scale,
                           min_obj_conf, min_frag_rel_conf,
                           only_annotated_objs=True):
  """Same contract as corresp.py:9-33 with the model store unpacked into
  ``obj_ids`` (model_store.dp_model['obj_ids']), ``frag_centers`` and
  ``frag_sizes`` (dicts obj_id -> f64[F,3] / f64[F]). project_to_surface is not
  covered (default off, infer.py:59-61)."""
  obj_confs = np.asarray(obj_confs, np.float32)
  frag_confs = np.asarray(frag_confs, np.float32)
  frag_coords = np.asarray(frag_coords, np.float32)
  h, w = obj_confs.shape[:2]
  tau_a = np.float32(min_obj_conf)
  tau_b = np.float32(min_frag_rel_conf)
  scale = 1.0 / output_scale                      # python double, corresp.py:56-57
  corresp = {}
  for obj_id in obj_ids:
    if only_annotated_objs and obj_id not in gt_obj_ids:
      continue
    conf_map = obj_confs[:, :, obj_id]
    mask = conf_map > tau_a
    if not mask.any():
      continue                                    # key absent, corresp.py:49
    centers = np.asarray(frag_centers[obj_id], np.float64)
    sizes = np.asarray(frag_sizes[obj_id], np.float64)
    px_id, frag_id, c2d, c3d, conf, conf_o, conf_f = [], [], [], [], [], [], []
    n_px = 0
    ys, xs = np.nonzero(mask)                     # raster order
    for y, x in zip(ys, xs):
      fc = frag_confs[y, x, obj_id - 1, :]
      thr = np.float32(fc.max() * tau_b)          # f32 * f32 -> f32
      for f in np.nonzero(fc > thr)[0]:
        loc = frag_coords[y, x, obj_id - 1, f, :]
        local = (loc.astype(np.float64) * sizes[f]).astype(np.float32)
        px_id.append(n_px)
        frag_id.append(int(f))
        c2d.append((scale * (np.float64(x) + 0.5), scale * (np.float64(y) + 0.5)))
        c3d.append(centers[f] + local.astype(np.float64))
        conf_o.append(conf_map[y, x])
        conf_f.append(fc[f])
        conf.append(np.float32(conf_map[y, x] * fc[f]))
      n_px += 1
    corresp[obj_id] = {
        'px_id': np.asarray(px_id, np.int64),
        'frag_id': np.asarray(frag_id, np.int64),
        'coord_2d': np.asarray(c2d, np.float64).reshape(-1, 2),
        'coord_3d': np.asarray(c3d, np.float64).reshape(-1, 3),
        'conf': np.asarray(conf, np.float32),
        'conf_obj': np.asarray(conf_o, np.float32),
        'conf_frag': np.asarray(conf_f, np.float32),
    }
  return corresp
