"""CPU oracle for the EPOS inference hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product. Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import, call, link or execute anything in here, and only as the checker.
The product (``epos_amd/``) never imports this package and fails loudly when
its HIP library is missing.

Pieces (each module cites the reference file:line it restates):

* ``net_ref``      -- torch-CPU fp32 restatement of ``epos_lib/model.py::predict``
                      (DeepLabv3+/Xception-65). TensorFlow 1.12 is not installable
                      here, so beyond the slim known-answer tests this part is
                      **parity unpinned** (see DESIGN.md).
* ``corresp_ref``  -- numpy restatement of ``epos_lib/corresp.py``; pinned by
                      golden vectors generated from the imported reference
                      (``tests/golden/make_golden.py``).
* ``fragment_ref`` -- numpy restatement of ``epos_lib/fragment.py``; pinned likewise.
* ``pnp_ref.c``    -- plain-C restatement of the pose-fitting stage. The reference
                      calls the un-vendored ``danini/progressive-x`` (branch
                      ``version-epos``, commit not recorded in the tree); its source
                      is absent, so this part is **parity unpinned** and validated
                      against synthetic known poses instead.
* ``epnp_ref.c``   -- plain-C restatement of ``cv2.solvePnPRansac(..., SOLVEPNP_EPNP)``
                      (scripts/infer.py:505-528). OpenCV is not installed here:
                      **parity unpinned**; checked against an independent numpy
                      statement of EPnP and of the RANSAC loop.
"""
