"""Synthetic workloads: seeded images, camera, targets and a model
store shaped like the reference's ObjectModelStore (datagen.py:24-154). There is no
network access for YCB-V images or the released checkpoints, so the benchmark and
the smoke test run on these."""
import numpy as np

YCBV_K = np.array([[1066.8, 0.0, 313.0], [0.0, 1067.5, 241.3], [0.0, 0.0, 1.0]])


class ModelStore(object):
  """Duck-typed stand-in for datagen.ObjectModelStore: dp_model['obj_ids'],
  frag_centers {obj_id: f64[F,3]}, frag_sizes {obj_id: f64[F]} (mm)."""

  def __init__(self, num_objs, num_frags, seed=0):
    rng = np.random.RandomState(seed)
    self.dp_model = {'obj_ids': list(range(1, num_objs + 1))}
    self.frag_centers, self.frag_sizes = {}, {}
    for o in self.dp_model['obj_ids']:
      radii = rng.uniform(30, 80, 3)
      d = rng.standard_normal((num_frags, 3))
      self.frag_centers[o] = d / np.linalg.norm(d, axis=1, keepdims=True) * radii
      self.frag_sizes[o] = rng.uniform(5, 40, num_frags)


def image(index, height, width, rank=0):
  """np.random.RandomState(1000*rank + i).randint(0, 256, (H, W, 3))."""
  rng = np.random.RandomState(1000 * rank + index)
  return rng.randint(0, 256, (height, width, 3)).astype(np.float32)


def targets(index, num_objs, objs_per_image=5, rank=0):
  """Localization targets: objs_per_image random objects x 1 instance (C2/C3)."""
  rng = np.random.RandomState(7919 * (1000 * rank + index) + 13)
  ids = rng.choice(np.arange(1, num_objs + 1), size=min(objs_per_image, num_objs),
                   replace=False)
  return {int(o): 1 for o in sorted(ids)}


def calibrate_logits(ckpt, decoder_features, std_obj=2.0, std_frag=3.0,
                     std_loc=0.3):
  """Rescales the random-init logits layers of ``ckpt`` (in place) so that, on the
  given decoder features [P, 256] (from a forward pass on a synthetic frame), the
  logits are zero-mean per channel with the requested standard deviations.

  Why: with the reference's initialisers (logits std 0.01, model.py:437) every
  confidence stays below tau_a = 0.1, and with any global rescale one class wins
  at every pixel, because random-init features of a noise image are dominated by
  their spatial mean. Either way the correspondence and RANSAC stages would get
  no work. After calibration each object passes tau_a on ~10 % of the pixels and
  keeps a few fragments per pixel (YCB-V-like correspondence counts); the network
  arithmetic is unchanged."""
  x = np.asarray(decoder_features, np.float64).reshape(-1, 256)
  mu = x.mean(axis=0)
  xc = x - mu
  for name, std in (('pred_obj_conf', std_obj), ('pred_frag_conf', std_frag),
                    ('pred_frag_loc', std_loc)):
    w = ckpt['logits/%s/weights' % name].reshape(256, -1).astype(np.float64)
    s = (xc[::7] @ w).std(axis=0) + 1e-12
    w = w * (std / s)
    ckpt['logits/%s/weights' % name] = w.astype(np.float32).reshape(1, 1, 256, -1)
    ckpt['logits/%s/biases' % name] = (-(mu @ w)).astype(np.float32)
  return ckpt
