"""``--vis`` of the inference script (scripts/infer.py:150-291, epos_lib/vis.py) without
the OSMesa renderer: the same grid of tiles per image -- input, ground-truth / estimated
pose overlays, ground-truth / predicted object labels, optional per-class confidences --
and the predicted fragment-field images (centres, local coordinates, reconstruction).

What differs from the reference on purpose: object poses are drawn by a small z-buffered
point-splat rasteriser in numpy over the model's vertices (``store.models[obj]['pts']``
when the meshes are loaded, the fragment centres otherwise) plus the object's coordinate
frame, instead of bop_renderer's shaded mesh (no OpenGL / OSMesa in this stack); label
colours come from a generated palette, not the ADE20K table. Host-side numpy + PIL only;
the dense head tensors are copied to the host for it, as the reference does
(infer.py:373).
"""
import colorsys
import math
import os

import numpy as np

TILE_SIZE = (300, 225)      # (width, height) of a grid tile, infer.py:170
VIS_EXT = 'jpg'


def palette(n=256):
  """n visually distinct colours; index 0 (background) is black."""
  out = np.zeros((n, 3), np.uint8)
  for i in range(1, n):
    h = (i * 0.61803398875) % 1.0
    s = 0.55 + 0.45 * ((i * 7) % 3) / 2.0
    v = 0.65 + 0.35 * ((i * 5) % 4) / 3.0
    out[i] = [int(255 * c) for c in colorsys.hsv_to_rgb(h, s, v)]
  return out


_PALETTE = palette()


def colorize_label_map(label):
  """[H,W] integer labels -> [H,W,3] uint8 (vis.py:79-96)."""
  label = np.asarray(label)
  if label.ndim != 2:
    raise ValueError('Expect 2-D input label. Got {}'.format(label.shape))
  return _PALETTE[np.mod(label, len(_PALETTE))]


def colorize_xyz(xyz):
  """3D points -> colours by their place in the bounding RGB box (vis.py:99-108)."""
  xyz = np.asarray(xyz, np.float64)
  v = xyz - xyz.min()
  m = v.max()
  return (255 * v / m if m > 0 else np.zeros_like(v)).astype(np.uint8)


def build_grid(tiles, tile_size, grid_rows=None, grid_cols=None):
  """Tiles of (width, height) = tile_size, row-major, into one image (vis.py:53-76)."""
  if not tiles:
    return np.zeros((tile_size[1], tile_size[0], 3), np.uint8)
  if grid_rows is None or grid_cols is None:
    grid_rows = max(1, int(math.sqrt(len(tiles))))
    grid_cols = int(math.ceil(len(tiles) / float(grid_rows)))
  w, h = tile_size
  grid = np.zeros((grid_rows * h, grid_cols * w, 3), np.uint8)
  for i, tile in enumerate(tiles):
    if tile.shape[:2] != (h, w):
      raise ValueError('tile %d is %s, expected %s' % (i, tile.shape[:2], (h, w)))
    r, c = divmod(i, grid_cols)
    grid[r * h:(r + 1) * h, c * w:(c + 1) * w] = tile[..., :3]
  return grid


def resize(im, size_wh):
  from PIL import Image
  im = np.asarray(im)
  if im.dtype != np.uint8:
    im = np.clip(im, 0, 255).astype(np.uint8)
  return np.asarray(Image.fromarray(im).resize(tuple(size_wh), Image.BILINEAR))


def write_text(im, text, color=(204, 204, 204)):
  from PIL import Image, ImageDraw
  pil = Image.fromarray(np.ascontiguousarray(im, np.uint8))
  ImageDraw.Draw(pil).text((4, 3), str(text), fill=tuple(color))
  return np.asarray(pil)


def project(pts, K, R, t):
  """[n,3] model points -> ([n,2] pixels, [n] depths)."""
  Y = np.asarray(pts, np.float64) @ np.asarray(R, np.float64).T + np.asarray(
      t, np.float64).reshape(1, 3)
  z = Y[:, 2]
  p = Y @ np.asarray(K, np.float64).T
  with np.errstate(divide='ignore', invalid='ignore'):
    uv = p[:, :2] / p[:, 2:3]
  return uv, z


def draw_coordinate_frame(im, K, R, t, size_px=15):
  """X/Y/Z axes of the object frame in red/green/blue (vis.py:111-138)."""
  from PIL import Image, ImageDraw
  f = 0.5 * (K[0][0] + K[1][1])
  a = 500.0 * size_px / f
  uv, z = project(np.array([[0., 0, 0], [a, 0, 0], [0, a, 0], [0, 0, a]]), K, R, t)
  pil = Image.fromarray(np.ascontiguousarray(im, np.uint8))
  if (z > 0).all() and np.isfinite(uv).all():
    d = ImageDraw.Draw(pil)
    for i in range(1, 4):
      col = [0, 0, 0]
      col[i - 1] = 255
      d.line([tuple(map(int, uv[0])), tuple(map(int, uv[i]))], fill=tuple(col), width=2)
  return np.asarray(pil)


def model_points(store, obj_id):
  models = getattr(store, 'models', None)
  if models and obj_id in models:
    return np.asarray(models[obj_id]['pts'], np.float64)
  return np.asarray(store.frag_centers[obj_id], np.float64)


def overlay_object_poses(rgb, K, poses, store, splat=1):
  """Estimated (or ground-truth) poses on top of the image: each object's model points,
  coloured by their object-frame position, z-buffered per pixel, blended 0.3 / 0.7 with
  the image like vis.visualize_object_poses (vis.py:141-176), plus the object frame."""
  rgb = np.asarray(rgb)
  h, w = rgb.shape[:2]
  ren = np.zeros((h, w, 3), np.float32)
  zbuf = np.full((h, w), np.inf)
  for pose in poses:
    if pose['obj_id'] not in store.frag_centers:
      continue
    pts = model_points(store, pose['obj_id'])
    if len(pts) > 20000:
      pts = pts[:: len(pts) // 20000 + 1]
    col = colorize_xyz(pts).astype(np.float32)
    uv, z = project(pts, K, pose['R'], pose['t'])
    ok = (z > 0) & np.isfinite(uv).all(1)
    u = np.round(uv[ok, 0]).astype(np.int64)
    v = np.round(uv[ok, 1]).astype(np.int64)
    zz, cc = z[ok], col[ok]
    order = np.argsort(-zz)                      # far first: the nearest write wins
    for dy in range(-splat, splat + 1):
      for dx in range(-splat, splat + 1):
        uu, vv = u[order] + dx, v[order] + dy
        m = (uu >= 0) & (uu < w) & (vv >= 0) & (vv < h)
        uu, vv, zo, co = uu[m], vv[m], zz[order][m], cc[order][m]
        closer = zo <= zbuf[vv, uu]
        zbuf[vv[closer], uu[closer]] = zo[closer]
        ren[vv[closer], uu[closer]] = co[closer]
  out = np.clip(0.3 * rgb.astype(np.float32) + 0.7 * ren, 0, 255).astype(np.uint8)
  for pose in poses:
    out = draw_coordinate_frame(out, K, pose['R'], pose['t'])
  return out


def visualize_pred_frag(frag_confs, frag_coords, output_size, store, vis_prefix, vis_dir,
                        vis_ext=VIS_EXT):
  """Per object: the most confident fragment of every pixel -> its centre, the predicted
  local coordinates scaled by the fragment size, and their sum (the 3D reconstruction),
  each as an RGB-box image; three grids (vis.py:251-319). frag_confs [h,w,O,F],
  frag_coords [h,w,O,F,3]; objects without fragments in the store are skipped."""
  from PIL import Image
  frag_confs, frag_coords = np.asarray(frag_confs), np.asarray(frag_coords)
  h, w, num_objs, num_frags = frag_confs.shape
  tiles = {'centers': [], 'coords': [], 'reconst': []}
  for obj_id in range(1, num_objs + 1):
    if obj_id not in store.frag_centers:
      continue
    top = np.argmax(frag_confs[:, :, obj_id - 1, :], axis=2).ravel()
    centers = np.asarray(store.frag_centers[obj_id])[top]
    rel = frag_coords[:, :, obj_id - 1].reshape(-1, num_frags, 3)[np.arange(top.size), top]
    coords = rel * np.asarray(store.frag_sizes[obj_id])[top][:, None]
    for key, val in (('centers', centers), ('coords', coords), ('reconst', centers + coords)):
      tile = colorize_xyz(val).reshape(h, w, 3)
      tile = resize(tile, output_size)
      tiles[key].append(write_text(tile, 'cls %d' % obj_id, (255, 255, 255)))
  os.makedirs(vis_dir, exist_ok=True)
  paths = []
  for key, lst in tiles.items():
    path = os.path.join(vis_dir, '%s_pred_frag_%s.%s' % (vis_prefix, key, vis_ext))
    Image.fromarray(build_grid(lst, output_size)).save(path)
    paths.append(path)
  return paths


def visualize(rgb, K, predictions, pred_poses, im_ind, store, vis_dir, gt_poses=None,
              gt_obj_label=None, flags=None):
  """One image's visualisations (infer.py:150-291). predictions: host arrays of ONE image
  -- pred_obj_label [h,w], pred_obj_conf [h,w,O+1], pred_frag_conf [h,w,O,F],
  pred_frag_loc [h,w,O,F,3]. Returns the paths written."""
  from PIL import Image
  fl = {'vis_gt_poses': True, 'vis_pred_poses': True, 'vis_gt_obj_labels': True,
        'vis_pred_obj_labels': True, 'vis_pred_obj_confs': False,
        'vis_gt_frag_fields': False, 'vis_pred_frag_fields': False}
  fl.update(flags or {})
  rgb = np.clip(np.asarray(rgb), 0, 255).astype(np.uint8)
  prefix = '%06d' % im_ind
  tiles = [write_text(resize(rgb, TILE_SIZE), 'input')]
  if fl['vis_gt_poses'] and gt_poses:
    tiles.append(write_text(resize(overlay_object_poses(rgb, K, gt_poses, store),
                                   TILE_SIZE), 'gt poses'))
  if fl['vis_pred_poses']:
    tiles.append(write_text(resize(overlay_object_poses(rgb, K, pred_poses, store),
                                   TILE_SIZE), 'pred poses'))
  if fl['vis_gt_obj_labels'] and gt_obj_label is not None:
    tiles.append(write_text(resize(colorize_label_map(gt_obj_label), TILE_SIZE),
                            'gt obj labels'))
  if fl['vis_pred_obj_labels']:
    tiles.append(write_text(resize(colorize_label_map(predictions['pred_obj_label']),
                                   TILE_SIZE), 'predicted obj labels'))
  if fl['vis_pred_obj_confs']:
    conf = np.asarray(predictions['pred_obj_conf'])
    for c in range(conf.shape[-1]):
      g = resize((255.0 * conf[:, :, c]).astype(np.uint8), TILE_SIZE)
      tiles.append(write_text(np.dstack([g, g, g]), 'cls %d' % c))
  os.makedirs(vis_dir, exist_ok=True)
  paths = []
  if fl['vis_pred_frag_fields']:
    hh, ww = np.asarray(predictions['pred_obj_label']).shape
    paths += visualize_pred_frag(predictions['pred_frag_conf'], predictions['pred_frag_loc'],
                                 (ww, hh), store, prefix, vis_dir)
  path = os.path.join(vis_dir, '%s_grid.%s' % (prefix, VIS_EXT))
  Image.fromarray(build_grid(tiles, TILE_SIZE)).save(path)
  return paths + [path]
