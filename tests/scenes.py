"""Synthetic frames for parity tests and golden vectors (pure numpy, deterministic).

Two generators:
  * `uniform_frame`  -- the benchmark's frame recipe: depth ~ U(0.6, 4.9) with 2 % of
    pixels at exactly the depth-range sentinels, logits ~ 3*N(0,1).
  * `room_frame`     -- an analytic ray-cast of a box-shaped room with one goal object,
    so that map cells are contiguous (exercises erosion/dilation, hit/view counters and
    the found decision), with speckle noise in the labels.
Poses follow the reference's sensor format [x, y, z, qw, qx, qy, qz] (task.py:190-193).
"""
from __future__ import annotations

import math

import numpy as np

CAM_HEIGHT = 1.096       # cfg/h_config.yaml:52
CAM_PITCH = -0.349066    # cfg/h_config.yaml:53

# model-class -> schema-id tables (infer.py:69-76 for 37/40-class EMSANet-style heads,
# infer.py:278-292 for the 150-class ADE head); every other class maps to 2 (occupied)
EMSANET_TABLE = {1: 1, 4: 3, 3: 4, 32: 6, 24: 7, 5: 8}
ADE_TABLE = {3: 1, 0: 2, 19: 3, 30: 3, 75: 3, 7: 4, 17: 5, 65: 6, 143: 7, 130: 7, 141: 7, 89: 7, 23: 8}


def schema_lut(n_classes: int) -> np.ndarray:
    table = ADE_TABLE if n_classes == 150 else EMSANET_TABLE
    lut = np.full(n_classes, 2, dtype=np.int64)
    for k, v in table.items():
        if k < n_classes:
            lut[k] = v
    return lut


def small_lut(n_classes: int) -> np.ndarray:
    """LUT for tiny class counts (tests with C=5..12): class k -> schema id k (capped at 8)."""
    return np.minimum(np.arange(n_classes), 8).astype(np.int64)


def _qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return (aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw)


def make_pose(x, floor_y, z, yaw, pitch=CAM_PITCH, cam_height=CAM_HEIGHT, agent_lift=0.1):
    """Agent standing at (x, z) with heading `yaw` about +y; camera `cam_height` above the
    agent, pitched by `pitch` about its local x axis.  Returns (agent7, camera7) float64."""
    q_yaw = (math.cos(yaw / 2), 0.0, math.sin(yaw / 2), 0.0)
    q_pitch = (math.cos(pitch / 2), math.sin(pitch / 2), 0.0, 0.0)
    q_cam = _qmul(q_yaw, q_pitch)
    agent = np.array([x, floor_y + agent_lift, z, *q_yaw], dtype=np.float64)
    camera = np.array([x, floor_y + agent_lift + cam_height, z, *q_cam], dtype=np.float64)
    return agent, camera


def _rot_from_pose(camera7):
    """Rotation the mapper applies to camera-frame points: R(q_cam * rotvec(pi,0,0))."""
    q = tuple(camera7[3:7])
    flip = (math.cos(math.pi / 2), math.sin(math.pi / 2), 0.0, 0.0)
    w, x, y, z = _qmul(q, flip)
    n = w * w + x * x + y * y + z * z
    return np.array([
        [1 - 2 * (y * y + z * z) / n, 2 * (x * y - z * w) / n, 2 * (x * z + y * w) / n],
        [2 * (x * y + z * w) / n, 1 - 2 * (x * x + z * z) / n, 2 * (y * z - x * w) / n],
        [2 * (x * z - y * w) / n, 2 * (y * z + x * w) / n, 1 - 2 * (x * x + y * y) / n]])


def softmax_stats(logits, temperature, lut):
    """Stage-1 outputs from logits (numpy float32; a plain restatement used to build
    inputs -- the checked stage-1 oracle lives in oracle/stage1_oracle.py)."""
    z = logits.astype(np.float32) / np.float32(temperature)
    z = z - z.max(axis=0, keepdims=True)
    e = np.exp(z)
    p = (e / e.sum(axis=0, keepdims=True)).astype(np.float32)
    c = p.shape[0]
    unc = (-(p * np.log(p + np.float32(1e-6))).sum(axis=0) / np.float32(np.log(c))).astype(np.float32)
    arg = p.argmax(axis=0)
    sem = lut[arg].astype(np.int32)
    maxp = p.max(axis=0).astype(np.float32)
    for gid in range(3, 9):
        members = [k for k in range(c) if lut[k] == gid]
        if members:
            pooled = p[members].sum(axis=0)
            maxp = np.where(sem == gid, pooled, maxp).astype(np.float32)
    return p, sem, unc, maxp


def uniform_frame(rng, h, w, n_classes, temperature, depth_range=(0.5, 5.0), lut=None,
                  sentinel_frac=0.02, with_probs=True):
    depth = rng.uniform(0.6, 4.9, (h, w)).astype(np.float32)
    bad = rng.random((h, w)) < sentinel_frac
    which = rng.random((h, w)) < 0.5
    depth[bad & which] = np.float32(depth_range[0])
    depth[bad & ~which] = np.float32(depth_range[1])
    frame = {"depth": depth, "semantic_gt": rng.integers(0, 9, (h, w)).astype(np.int32)}
    if n_classes:
        logits = (3.0 * rng.standard_normal((n_classes, h, w))).astype(np.float32)
        frame["logits"] = logits
        if with_probs:
            lut = schema_lut(n_classes) if lut is None else lut
            p, sem, unc, maxp = softmax_stats(logits, temperature, lut)
            frame.update(semantic_prob=p, semantic=sem, semantic_unc=unc, semantic_max_prob=maxp)
    else:
        frame["semantic"] = rng.integers(0, 9, (h, w)).astype(np.int32)
    return frame


def _ray_box(origin, dirs, lo, hi):
    """Slab test: distance t >= 0 along each ray to the box [lo, hi], inf if missed."""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / dirs
        t0 = (lo - origin) * inv
        t1 = (hi - origin) * inv
    tmin = np.nanmax(np.minimum(t0, t1), axis=-1)
    tmax = np.nanmin(np.maximum(t0, t1), axis=-1)
    hit = (tmax >= np.maximum(tmin, 0.0))
    return np.where(hit & (tmin > 0), tmin, np.inf)


def room_frame(rng, h, w, n_classes, temperature, camera7, *, hfov_deg=59.0, room=(0.0, 0.0, 4.5, 4.5),
               floor_y=0.0, wall_out=0.12, box=(2.6, 2.6, 3.3, 3.2, 0.7), goal_schema=3,
               depth_range=(0.5, 5.0), lut=None, speckle=0.03, sharp=6.0, quantum=None):
    """Ray-cast a room [x0,z0,x1,z1] whose walls stand `wall_out` outside the navigable
    bounds, with one goal-class box (x0,z0,x1,z1,height).  Depth is the z-depth the
    mapper's pinhole model expects.  Labels: floor 1, wall 2, box `goal_schema`, with a
    `speckle` fraction of pixels relabelled at random (goal class included)."""
    hfov = np.deg2rad(hfov_deg)
    fx = w / (2.0 * np.tan(hfov / 2.0))
    fy = h / (2.0 * np.tan((h / w * hfov) / 2.0))
    cc, rr = np.meshgrid(np.arange(w), np.arange(h))
    dirs_cam = np.stack([(cc - w / 2) / fx, (rr - h / 2) / fy, np.ones((h, w))], axis=-1)
    rot = _rot_from_pose(camera7)
    dirs = dirs_cam @ rot.T
    origin = np.asarray(camera7[:3], dtype=np.float64)
    x0, z0, x1, z1 = room
    with np.errstate(divide="ignore", invalid="ignore"):
        t_floor = (floor_y - origin[1]) / dirs[..., 1]
        t_walls = np.stack([(x0 - wall_out - origin[0]) / dirs[..., 0],
                            (x1 + wall_out - origin[0]) / dirs[..., 0],
                            (z0 - wall_out - origin[2]) / dirs[..., 2],
                            (z1 + wall_out - origin[2]) / dirs[..., 2]], axis=-1)
    t_floor = np.where(t_floor > 0, t_floor, np.inf)
    t_wall = np.where(t_walls > 0, t_walls, np.inf).min(axis=-1)
    bx0, bz0, bx1, bz1, bh = box
    t_box = _ray_box(origin, dirs, np.array([bx0, floor_y, bz0]), np.array([bx1, floor_y + bh, bz1]))
    t = np.minimum(np.minimum(t_floor, t_wall), t_box)
    label = np.where(t == t_box, goal_schema, np.where(t == t_floor, 1, 2)).astype(np.int32)
    depth = np.clip(t, depth_range[0], depth_range[1]).astype(np.float32)  # dirs_cam z == 1
    if quantum:
        depth = (np.round(depth / quantum) * quantum).astype(np.float32)
    noise = rng.random((h, w)) < speckle
    label = np.where(noise, rng.integers(0, 9, (h, w)), label).astype(np.int32)
    frame = {"depth": depth, "semantic_gt": label.copy()}
    if n_classes:
        lut = schema_lut(n_classes) if lut is None else lut
        # pick, per schema id, the first model class that maps to it
        rep = np.array([int(np.argmax(lut == s)) if np.any(lut == s) else 0 for s in range(9)])
        logits = rng.standard_normal((n_classes, h, w)).astype(np.float32)
        boost = (sharp * rng.random((h, w))).astype(np.float32)
        np.put_along_axis(logits, rep[label][None], logits[rep[label], rr, cc][None] + boost[None], axis=0)
        p, sem, unc, maxp = softmax_stats(logits, temperature, lut)
        frame.update(logits=logits, semantic_prob=p, semantic=sem, semantic_unc=unc,
                     semantic_max_prob=maxp)
    else:
        frame["semantic"] = label.copy()
    return frame


def make_obs(frame, agent7, camera7, goal_index=0):
    obs = {k: v.copy() for k, v in frame.items() if k != "logits"}
    obs["agent_position_and_rotation"] = np.asarray(agent7, dtype=np.float64).copy()
    obs["camera_position_and_rotation"] = np.asarray(camera7, dtype=np.float64).copy()
    obs["objectgoal"] = np.array([goal_index])
    return obs
