"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's per-episode detection metric,
`SemanticMapper.compute_fp_fn_detections` (sem_objnav/obj_nav/mapper.py:1053-1112).

Only tests/ may import this module; the product is `semantic_search_amd.SemanticMapper
.compute_fp_fn_detections`.  Pinned against tests/golden/metrics/fpfn.npz, which holds outputs of the
reference method itself (tests/golden/make_fpfn_golden.py); the 3x3 dilation (cv2.dilate,
mapper.py:1105-1108) is restated in oracle/raster.py and is the one unpinned seam.
"""
from __future__ import annotations

import numpy as np
import scipy.ndimage

from oracle import raster


def box_cells(center, sizes, lower, pixel2meter, shape):
    """Half-open cell rectangle of one axis-aligned box (mapper.py:1063-1072): world2map of the
    two corners, each index clipped to the map."""
    def cell(p):
        d = np.asarray(p, dtype=np.float64) - np.asarray(lower, dtype=np.float64)
        return np.floor(np.array([d[2], d[0]]) / pixel2meter).astype(np.int64)   # mapper.py:851-863
    p1, p2 = cell(np.asarray(center) - np.asarray(sizes)), cell(np.asarray(center) + np.asarray(sizes))
    r1, c1 = np.clip(p1[0], 0, shape[0] - 1), np.clip(p1[1], 0, shape[1] - 1)
    r2, c2 = np.clip(p2[0], 0, shape[0] - 1), np.clip(p2[1], 0, shape[1] - 1)
    return int(r1), int(r2), int(c1), int(c2)


def fp_fn_detections(semantic_map, gt_map, goal_sem_id, centers, sizes, lower, pixel2meter=0.03,
                     hits=None, views=None):
    """Returns (fp, fn, is_fn_bbox).  `hits`/`views` given <=> a hit/view rule (mapper.py:1054)."""
    shape = semantic_map.shape
    pred = semantic_map == goal_sem_id                                   # mapper.py:1060
    all_boxes = np.zeros(shape, dtype=bool)
    fn, is_fn = 0, []
    for center, size in zip(centers, sizes):
        r1, r2, c1, c2 = box_cells(center, size, lower, pixel2meter, shape)
        if not np.any(gt_map[r1:r2, c1:c2] == goal_sem_id):             # mapper.py:1073-1084
            is_fn.append(False)
            continue
        all_boxes[r1:r2, c1:c2] = True
        if hits is not None:                                             # mapper.py:1085-1093
            seen = np.any((hits[r1:r2, c1:c2] > 0) & (views[r1:r2, c1:c2] != np.inf))
        else:                                                            # mapper.py:1094-1100
            seen = np.any(pred[r1:r2, c1:c2])
        is_fn.append(not seen)
        fn += 0 if seen else 1
    fp_mask = (~all_boxes & pred).astype(np.uint8)                       # mapper.py:1101
    k = int(0.1 / pixel2meter)                                           # mapper.py:1104
    fp_mask = raster.dilate_max(fp_mask, k, k)                           # mapper.py:1105-1108
    _, fp = scipy.ndimage.label(fp_mask, scipy.ndimage.generate_binary_structure(2, 2))   # mapper.py:1109-1111
    return int(fp), fn, is_fn
