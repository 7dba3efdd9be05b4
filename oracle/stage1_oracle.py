"""TEST INFRASTRUCTURE ONLY -- CPU oracle for stage 1 (softmax / entropy / confidence).

Restates the tail of the reference's segmentation wrappers with torch CPU tensors:
  EmsanetWrapper.predict   sem_objnav/segment/infer.py:97-118
  SegformerWrapper.predict sem_objnav/segment/infer.py:302-327
(both do: logits / T -> softmax over the class axis -> normalised entropy with a 1e-6
epsilon -> max / argmax -> schema LUT -> goal-class probability pooling).

Pinned: tests/golden/stage1/*.npz were recorded from those two predict() methods themselves (the
reference's infer.py executed in the build container on synthetic network outputs, its absent
model dependencies replaced by inert import shims: tests/golden/make_stage1_golden.py);
tests/test_stage1_golden.py checks that this file reproduces them bit for bit.
"""
from __future__ import annotations

import numpy as np
import torch


def pooling_groups(lut, table_order=None):
    """For schema goal ids 3..8: the model classes mapped to that id, in the order the
    reference's objnav2pred dict lists them (insertion order of pred2objnav_occ_id,
    infer.py:77-79, 293-295).  `table_order` is that insertion order; default ascending."""
    lut = np.asarray(lut)
    order = list(range(len(lut))) if table_order is None else list(table_order)
    return {g: [k for k in order if lut[k] == g] for g in range(3, 9)}


def predict_tail(logits: np.ndarray, temperature: float, lut, table_order=None, pool_groups=None):
    """logits (C,H,W) float32 -> dict with the reference's predict() keys.  `pool_groups`
    ({goal id: model classes}) stands for objnav2pred when it is not the table's inverse."""
    lut = np.asarray(lut)
    with torch.no_grad():
        z = torch.from_numpy(np.ascontiguousarray(logits)) / temperature
        prob = torch.nn.functional.softmax(z, dim=0)
        unc = -torch.sum(prob * torch.log(prob + 1e-6), dim=0) / np.log(prob.shape[0])
        max_probs, arg = prob.max(dim=0)
    semantic = lut[arg.numpy()].astype(np.int32)
    max_probs = max_probs.numpy().copy()
    probs = prob.numpy()
    groups = pooling_groups(lut, table_order)
    if pool_groups is not None:
        groups = {g: list(pool_groups.get(g, [])) for g in range(3, 9)}
    for gid, members in groups.items():
        mask = semantic == gid
        max_probs[mask] = probs[members, :].sum(axis=0)[mask]
    return {"semantic": semantic, "semantic_max_prob": max_probs, "semantic_prob": probs,
            "semantic_unc": unc.numpy(), "argmax": arg.numpy()}
