"""ORACLE (test infrastructure, not product): numpy restatement of
``epos_lib/fragment.py::fragmentation_fps`` (fragment.py:8-54) and of the
fragment-size rule of ``epos_lib/datagen.py:110-124`` (max side of the fragment's
bounding box, floor 5 mm -- restated from SURVEY.md a17).

PINNED for the FPS part: golden vectors from the imported reference function
(tests/golden/make_golden.py). The reference uses a cKDTree for nearest-centre
queries; brute-force distances give the same answer up to exact ties.
"""
import numpy as np


def fragmentation_fps(vertices, num_frags):
  vertices = np.asarray(vertices, np.float64)
  # Distances to the origin (fragment.py:27-32: FPS is seeded with the origin).
  nn_dists = np.sqrt((vertices ** 2).sum(axis=1))
  centers = []
  for _ in range(num_frags):
    ind = int(np.argmax(nn_dists))                               # :36
    c = vertices[ind]
    centers.append(c)
    nn_dists[ind] = -1                                           # :41
    nn_dists = np.minimum(nn_dists,
                          np.linalg.norm(vertices - c, axis=1))  # :42-43
  centers = np.array(centers)                                    # origin dropped
  d2 = ((vertices[:, None, :] - centers[None, :, :]) ** 2).sum(axis=2)
  return centers, np.argmin(d2, axis=1)


def fragment_sizes(vertices, vertex_frag_ids, num_frags, min_size=5.0):
  sizes = np.zeros(num_frags, np.float64)
  for f in range(num_frags):
    pts = vertices[vertex_frag_ids == f]
    side = (pts.max(axis=0) - pts.min(axis=0)).max() if len(pts) else 0.0
    sizes[f] = max(side, min_size)
  return sizes
