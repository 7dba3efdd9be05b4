"""Golden vectors for `SemanticMapper.compute_fp_fn_detections` (mapper.py:1053-1112).

    python tests/golden/make_fpfn_golden.py     (build container only: imports the reference)

The reference class is imported like in make_golden.py (stand-ins for the absent third-party
modules; `cv2.dilate` / `getStructuringElement` are the restatements in thirdparty_seams.py, so
that one seam is unpinned).  Its maps are filled with synthetic blobs, the reference method is
called on synthetic bounding boxes, and inputs + results go to tests/golden/metrics/fpfn.npz.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import thirdparty_seams  # noqa: E402
from make_golden import load_reference  # noqa: E402

BOUNDS = np.array([0.5, -0.2, -1.0, 4.7, 2.5, 3.8])      # 160 rows (z) x 140 cols (x)
CASES = [
    dict(name="argmax", prob=True, rule="height_map_argmax", goal_index=0),
    dict(name="hv", prob=True, rule="height_map_hv", goal_index=2),
    dict(name="direct", prob=False, rule="height_map_argmax", goal_index=5),
    dict(name="prob_hv", prob=True, rule="height_map_prob_hv", goal_index=1),
]


def blobs(rng, shape, n, value, out, size=(3, 14)):
    for _ in range(n):
        r, c = rng.integers(0, out.shape[0]), rng.integers(0, out.shape[1])
        h, w = rng.integers(size[0], size[1], 2)
        out[r:r + h, c:c + w] = value


def make_boxes(rng, mapper, gt_goal_cells, n):
    """Boxes around ground-truth goal cells, random boxes, boxes hanging over the map edge."""
    boxes = []
    lower = mapper.map_bounds[:3]
    for k in range(n):
        kind = k % 4
        if kind < 2 and len(gt_goal_cells):
            # alternate between the upper (hit) and the lower (missed) half of the map
            half = gt_goal_cells[(gt_goal_cells[:, 0] >= 80) == (kind == 1)]
            pick = half if len(half) else gt_goal_cells
            r, c = pick[rng.integers(len(pick))]
            cz, cx = (r + 0.5) * 0.03 + lower[2], (c + 0.5) * 0.03 + lower[0]
        elif kind == 2:
            cx = rng.uniform(lower[0], mapper.map_bounds[3])
            cz = rng.uniform(lower[2], mapper.map_bounds[5])
        else:
            cx, cz = lower[0] + rng.uniform(-0.3, 0.2), mapper.map_bounds[5] + rng.uniform(-0.2, 0.3)
        sizes = np.array([rng.uniform(0.05, 0.5), rng.uniform(0.1, 0.6), rng.uniform(0.05, 0.5)])
        boxes.append(types.SimpleNamespace(center=np.array([cx, rng.uniform(0.2, 1.0), cz]), sizes=sizes))
    return boxes


def main():
    ref = load_reference()
    out = {}
    for ci, case in enumerate(CASES):
        rng = np.random.default_rng(100 + ci)
        m = ref.SemanticMapper(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0),
                               probablistic_mapping=case["prob"], prob_aggregate=case["rule"])
        m.reset(BOUNDS.copy())
        goal = int(m.goal2sem_id[np.array([case["goal_index"]])][0])
        shape = m.semantic_map.shape
        sem = rng.integers(0, 3, shape).astype(np.uint32)
        gt = rng.integers(0, 3, shape).astype(np.uint32)
        blobs(rng, shape, 5, goal, gt[:80])
        blobs(rng, shape, 5, goal, gt[80:])
        blobs(rng, shape, 6, goal + 1 if goal < 8 else 3, gt)
        blobs(rng, shape, 14, goal, sem, size=(1, 9))
        # upper half of the map: predictions overlap part of the ground truth (hit boxes);
        # lower half: no prediction anywhere near a ground-truth object (missed boxes)
        upper = np.arange(shape[0])[:, None] < shape[0] // 2
        sem[(gt == goal) & upper & (rng.random(shape) < 0.35)] = goal
        near_gt = thirdparty_seams.dilate((gt == goal).astype(np.uint8), np.ones((41, 41), np.uint8)) > 0
        sem[near_gt & ~upper & (sem == goal)] = 0
        m.gt_map[...] = gt
        m.semantic_map[...] = sem     # without probabilistic mapping gt_map IS semantic_map (mapper.py:257)
        sem, gt = m.semantic_map.copy(), m.gt_map.copy()
        if "hv" in case["rule"]:
            hits = np.where(rng.random(shape) < 0.08, rng.integers(1, 5, shape), 0).astype(np.float32)
            hits[near_gt & ~upper] = 0
            views = rng.integers(0, 9, shape).astype(np.float32)
            views[rng.random(shape) < 0.05] = np.inf
            m.hits[...] = hits
            m.views[...] = views
            out[f"{case['name']}_hits"] = hits
            out[f"{case['name']}_views"] = views
        boxes = make_boxes(rng, m, np.argwhere(gt == goal), 14)
        fp, fn, is_fn = m.compute_fp_fn_detections(np.array([case["goal_index"]]), boxes)
        pre = case["name"]
        out[f"{pre}_semantic_map"] = sem
        out[f"{pre}_gt_map"] = gt
        out[f"{pre}_centers"] = np.stack([b.center for b in boxes])
        out[f"{pre}_sizes"] = np.stack([b.sizes for b in boxes])
        out[f"{pre}_goal_index"] = np.int64(case["goal_index"])
        out[f"{pre}_prob"] = np.bool_(case["prob"])
        out[f"{pre}_rule"] = np.str_(case["rule"])
        out[f"{pre}_fp"] = np.int64(fp)
        out[f"{pre}_fn"] = np.int64(fn)
        out[f"{pre}_is_fn_bbox"] = np.array(is_fn, dtype=bool)
        print(f"{pre:8s} goal={goal} fp={fp} fn={fn} is_fn={np.array(is_fn, dtype=int)}")
    out["bounds"] = BOUNDS
    out["names"] = np.array([c["name"] for c in CASES])
    np.savez_compressed(os.path.join(HERE, "metrics", "fpfn.npz"), **out)


if __name__ == "__main__":
    main()
