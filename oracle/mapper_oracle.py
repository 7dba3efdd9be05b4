"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the semantic-map fusion hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (semantic-search_amd/) never does.

It restates, stage by stage, what the reference's `SemanticMapper.update_map`
(sem_objnav/obj_nav/mapper.py:276-849) and `reset` (mapper.py:193-269) compute, with
numpy for the integer / float32 map arithmetic and the C file oracle/geometry.c for the
float64 per-pixel geometry.  Pinning: tests/test_oracle_golden.py checks this module
against tests/golden/*.npz, which were recorded from the reference's own mapper.py (see
tests/golden/make_golden.py for how, and for which third-party seams are unpinned).

Not restated here: ego-map rendering (mapper.py:876-998; see egomap_oracle.py).  The path
overlay (mapper.py:822-833) is restated with a pluggable line raster (`line_fn`, default
oracle/raster.py line: an approximation of cv2.line, like the one the fixtures were recorded with).
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess

import numpy as np
import scipy.special
import scipy.stats

from . import raster

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle_geometry.so")
_lib = None

AVG_RULES = ("avg_prob", "w_avg_prob", "w2_avg_prob")
KNOWN_RULES = AVG_RULES + ("log_odds", "goal_decay", "argmax", "hv", "prob_hv", "stubborn")


def build_geometry_lib(force: bool = False) -> str:
    """Compile oracle/geometry.c with gcc (contraction off) into oracle/_build/."""
    src = os.path.join(_HERE, "geometry.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_LIB_PATH), exist_ok=True)
        subprocess.check_call(
            ["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", _LIB_PATH, src, "-lm"]
        )
    return _LIB_PATH


def _geometry():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build_geometry_lib()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_project.restype = None
    return _lib


# --------------------------------------------------------------------------- pose (a-3)
def quat_mul(a, b):
    """Hamilton product of (w, x, y, z) tuples, plain left-to-right evaluation."""
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return (
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
    )


def quat_from_rotvec(rx, ry, rz):
    """exp of the pure quaternion (0, r/2): numpy-quaternion from_rotation_vector."""
    x, y, z = rx / 2, ry / 2, rz / 2
    n = math.sqrt(x * x + y * y + z * z)
    if n > 1e-14:
        s = math.sin(n) / n
        e = math.exp(0.0)
        return (e * math.cos(n), e * s * x, e * s * y, e * s * z)
    return (math.exp(0.0), 0.0, 0.0, 0.0)


def quat_to_matrix(q):
    """numpy-quaternion as_rotation_matrix, array branch: every entry divided by the
    Cayley norm w^2+x^2+y^2+z^2."""
    w, x, y, z = (float(v) for v in q)
    n = w * w + x * x + y * y + z * z
    m = np.empty((3, 3), dtype=np.float64)
    m[0, 0] = 1.0 - 2 * (y ** 2 + z ** 2) / n
    m[0, 1] = 2 * (x * y - z * w) / n
    m[0, 2] = 2 * (x * z + y * w) / n
    m[1, 0] = 2 * (x * y + z * w) / n
    m[1, 1] = 1.0 - 2 * (x ** 2 + z ** 2) / n
    m[1, 2] = 2 * (y * z - x * w) / n
    m[2, 0] = 2 * (x * z - y * w) / n
    m[2, 1] = 2 * (y * z + x * w) / n
    m[2, 2] = 1.0 - 2 * (x ** 2 + y ** 2) / n
    return m


def camera_rotation_matrix(camera_pose7):
    """R(q_cam * rotvec(pi,0,0)) -- mapper.py:294-301."""
    q = tuple(float(v) for v in np.asarray(camera_pose7, dtype=np.float64)[3:7])
    return quat_to_matrix(quat_mul(q, quat_from_rotvec(math.pi, 0.0, 0.0)))


def focal_lengths(height, width, hfov_deg):
    """fx, fy of pointcloud() -- mapper.py:21-25 (vfov is linear in hfov)."""
    hfov = np.deg2rad(hfov_deg)
    fx = width / (2.0 * np.tan(hfov / 2.0))
    vfov = height / width * hfov
    fy = height / (2.0 * np.tan(vfov / 2.0))
    return float(fx), float(fy)


def cell_of(world_xyz, lower_xyz, pixel2meter):
    """world2map for one point -- mapper.py:851-863; returns (row, col) int64."""
    d = np.asarray(world_xyz, dtype=np.float64) - np.asarray(lower_xyz, dtype=np.float64)
    uv = np.stack([d[..., 2], d[..., 0]], axis=-1)
    return np.floor(uv / pixel2meter).astype(np.int64)


# --------------------------------------------------------------------------- stages 2+3
def project_frame(depth, hfov_deg, depth_range, camera_pose7, lower_xyz, pixel2meter):
    """Stages a-2, a-4..a-6 for one frame.  `depth` (H,W) float32 is modified in place.
    Returns rows, cols (int64), heights (float64), valid (bool), all flat (H*W,)."""
    assert depth.dtype == np.float32 and depth.flags.c_contiguous
    h, w = depth.shape
    n = h * w
    fx, fy = focal_lengths(h, w, hfov_deg)
    rot = np.ascontiguousarray(camera_rotation_matrix(camera_pose7))
    trans = np.ascontiguousarray(np.asarray(camera_pose7, dtype=np.float64)[:3])
    lower = np.ascontiguousarray(np.asarray(lower_xyz, dtype=np.float64))
    rows = np.empty(n, dtype=np.int64)
    cols = np.empty(n, dtype=np.int64)
    heights = np.empty(n, dtype=np.float64)
    valid = np.empty(n, dtype=np.uint8)
    p = ctypes.c_void_p
    _geometry().oracle_project(
        p(depth.ctypes.data), ctypes.c_int32(h), ctypes.c_int32(w),
        ctypes.c_float(np.float32(depth_range[0])), ctypes.c_float(np.float32(depth_range[1])),
        ctypes.c_double(fx), ctypes.c_double(fy),
        p(rot.ctypes.data), p(trans.ctypes.data), p(lower.ctypes.data),
        ctypes.c_double(pixel2meter),
        p(rows.ctypes.data), p(cols.ctypes.data), p(heights.ctypes.data), p(valid.ctypes.data),
    )
    return rows, cols, heights, valid.astype(bool)


def tallest_per_cell(rows, cols, heights):
    """a-8: one pixel per distinct (row, col): the tallest, ties -> lowest flat index
    (numpy_indexed group_by(...).argmax semantics, mapper.py:331)."""
    n = rows.shape[0]
    order = np.lexsort((np.arange(n), -heights, cols, rows))
    r, c = rows[order], cols[order]
    head = np.ones(n, dtype=bool)
    head[1:] = (r[1:] != r[:-1]) | (c[1:] != c[:-1])
    return order[head]


def select_update_pixels(rows, cols, heights, valid, cam_y, cutoff, map_h, map_w):
    """a-7/a-8: winners that also pass the height cut-off and the upper-bound test,
    ascending flat index (mapper.py:319-335)."""
    winners = tallest_per_cell(rows, cols, heights)
    keep = (heights[winners] < (cam_y + cutoff)) & valid[winners]
    keep &= (rows[winners] < map_h) & (cols[winners] < map_w)
    return np.sort(winners[keep])


# --------------------------------------------------------------------------- the mapper
class OracleMapper:
    """CPU statement of the reference mapper's state and per-frame update."""

    def __init__(self, *, num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0),
                 pixel2meter=0.03, height_cutoff_m=0.1, entropy_threshold=0.5,
                 probablistic_mapping=False, prob_aggregate="height_map_argmax",
                 success_dist_m=1.0, seg_model_n_classes=None, filter_goal_prob=None,
                 schema_lut=None, collect_nb_data=False, nb_models=None, objnav_to_sem_pred=None,
                 path_width_m=0.1, line_fn=None, path_overlay=True):
        self.num_semantic_classes = num_semantic_classes
        self.depth_hfov = depth_hfov
        self.depth_range = depth_range
        self.pixel2meter = pixel2meter
        self.height_cutoff_m = height_cutoff_m
        self.probablistic_mapping = probablistic_mapping
        self.prob_aggregate = prob_aggregate
        self.found_dist_m = success_dist_m
        self.sem_model_n_classes = seg_model_n_classes
        self.filter_goal_prob = filter_goal_prob
        self.schema_lut = None if schema_lut is None else np.asarray(schema_lut)
        self.collect_nb_data = collect_nb_data
        self.path_width_px = int(path_width_m / pixel2meter)
        self.path_id = num_semantic_classes + 1
        self.line_fn = line_fn or raster.line
        self.path_overlay = path_overlay   # False: the state after the found decision, as ssmap_step leaves it
        self.nb_models = nb_models                    # {goal id: dict(classes_, class_prior_, theta_, var_)}
        self.objnav_to_sem_pred = objnav_to_sem_pred  # {goal id: model classes of that goal}
        self.use_height = prob_aggregate.startswith("height_map_")
        self.rule = prob_aggregate[len("height_map_"):] if self.use_height else prob_aggregate
        if self.rule not in KNOWN_RULES:
            raise RuntimeError("Invalid prob_aggregate")
        # rule constants (mapper.py:145-169)
        self.entropy_threshold = entropy_threshold
        self.hv_limit_obj_view_dist, self.hv_min_views, self.hv_ratio = 0.0, 10, 0.5
        self.goal_decay, self.goal_mark = 0.9, 2.0
        if num_semantic_classes == 9:
            self.goal2sem_id = np.array([3, 4, 5, 6, 7, 8])
        elif num_semantic_classes in (41, 38):
            self.goal2sem_id = np.array([3, 11, 14, 18, 22, 10])
        self.color_pallete = palette(num_semantic_classes + 3)
        self.color_pallete[num_semantic_classes + 1] = np.array([128, 0, 128], dtype=np.uint8)
        self.last_update_pixels = None
        self.last_update_cells = None

    # ---- reset (a-19)
    def reset(self, map_bounds):
        b = np.asarray(map_bounds)
        self.map_bounds = b
        nr = int(np.ceil(np.abs(b[5] - b[2]) / self.pixel2meter).astype(np.int32))
        nc = int(np.ceil(np.abs(b[3] - b[0]) / self.pixel2meter).astype(np.int32))
        f32, u32 = np.float32, np.uint32
        self.semantic_map = np.zeros((nr, nc), u32)
        self.semantic_map_color = np.zeros((nr, nc, 3), np.uint8)
        self.height_map = np.full((nr, nc), -np.inf, f32)
        self.path_mask = np.zeros((nr, nc), np.uint8)
        self.prev_pos_px = None
        self.unc_map = None
        if self.collect_nb_data or (self.probablistic_mapping and self.rule == "stubborn"):
            # mapper.py:244-250, 260-265
            self.total_view = np.zeros((nr, nc), u32)
            self.cumulative_conf = np.zeros((nr, nc), f32)
            self.max_conf = np.zeros((nr, nc), f32)
            self.max_other_conf = np.zeros((nr, nc), f32)
            self.eps_nb_data = {"goal": [], "non_goal": [], "goal_id": None}
        if not self.probablistic_mapping:
            self.gt_map = self.semantic_map
            return
        self.gt_map = np.zeros((nr, nc), u32)
        c = self.sem_model_n_classes
        if self.rule == "goal_decay":
            self.goal_map = np.zeros((nr, nc), f32)
        elif self.rule == "log_odds":
            self.log_odds_seg_map = np.zeros((nr, nc, c), f32)
        elif self.rule in AVG_RULES:
            self.avg_prob_seg_map = np.zeros((nr, nc, c), f32)
            self.avg_prob_weight_map = np.zeros((nr, nc, 1), f32)
        elif self.rule in ("hv", "prob_hv"):
            if not self.use_height:
                raise RuntimeError("Invalid prob_aggregate")
            self.background_map = np.zeros((nr, nc), u32)
            self.hits = np.zeros((nr, nc), f32)
            self.views = np.zeros((nr, nc), f32)
        self.unc_map = np.full((nr, nc, 1), -0.5, f32)

    # ---- helpers
    def _schema(self, ids):
        return self.schema_lut[ids]

    def _occupancy_override(self, ids, cells):
        """a-16: ids 0/1/2 become occupied (2) / free (1) from the height map."""
        if not self.use_height:
            return
        h = self.height_map[cells]
        low = (ids == 0) | (ids == 1) | (ids == 2)
        occupied = (h > 0.1) & low
        free = (h != -np.inf) * (h < 0.1) & low
        if self.num_semantic_classes == 9:
            ids[occupied] = 2
            ids[free] = 1
        elif self.rule == "log_odds":
            raise RuntimeError("Invalid num_semantic_classes for height_map_log_odds aggregation")

    def _filter_goal(self, ids, conf, goal_id):
        if self.filter_goal_prob is not None:
            ids[(conf < self.filter_goal_prob) & (ids == goal_id)] = 0

    # ---- one frame
    def update(self, obs):
        depth = obs["depth"]
        agent = np.asarray(obs["agent_position_and_rotation"])
        camera = np.asarray(obs["camera_position_and_rotation"])
        lower = self.map_bounds[:3]
        goal_id = self.goal2sem_id[obs["objectgoal"]][0]
        map_h, map_w = self.semantic_map.shape

        rows, cols, heights, valid = project_frame(
            depth, self.depth_hfov, self.depth_range, camera, lower, self.pixel2meter)
        pix = select_update_pixels(rows, cols, heights, valid, camera[1],
                                   self.height_cutoff_m, map_h, map_w)
        cells = (rows[pix], cols[pix])
        self.last_update_pixels, self.last_update_cells = pix, cells

        self.height_map[cells] = np.maximum(self.height_map[cells], heights[pix] - agent[1])

        hv_rule = self.probablistic_mapping and self.rule in ("hv", "prob_hv")
        if not self.probablistic_mapping:
            ids = obs["semantic"].reshape(-1)[pix]
            if self.prob_aggregate == "height_map_argmax":
                self._occupancy_override(ids, cells)
        else:
            self.gt_map[cells] = obs["semantic_gt"].reshape(-1)[pix]
            fuse = getattr(self, "_fuse_" + self.rule)
            cells, ids = fuse(obs, pix, cells, goal_id, agent, lower)
        if not hv_rule:
            self.semantic_map[cells] = ids
        self.semantic_map_color[cells] = self.color_pallete[self.semantic_map[cells]]
        found = self._found(agent, lower, goal_id, hv_rule)
        # path overlay (mapper.py:822-839): drawn after the found decision, from the second frame on
        robot = cell_of(agent[:3], lower, self.pixel2meter)
        if self.prev_pos_px is not None and self.path_overlay:
            self.path_mask = self.line_fn(self.path_mask, (self.prev_pos_px[1], self.prev_pos_px[0]),
                                          (robot[1], robot[0]), 1, self.path_width_px)
            on = self.path_mask.astype(bool)
            self.semantic_map[on] = self.path_id
            self.semantic_map_color[on] = self.color_pallete[self.path_id]
        self.prev_pos_px = robot
        return found

    # ---- rules (stage 4)
    def _fuse_argmax(self, obs, pix, cells, goal_id, agent, lower):
        ids = obs["semantic"].reshape(-1)[pix]
        if "semantic_unc" in obs:
            self.unc_map[cells] = obs["semantic_unc"].reshape(-1, 1)[pix]
        if self.filter_goal_prob is not None:
            self._filter_goal(ids, 1 - obs["semantic_unc"].reshape(-1)[pix], goal_id)
        self._occupancy_override(ids, cells)
        if self.collect_nb_data or self.rule == "stubborn":
            self._goal_statistics(obs, pix, cells, ids, goal_id)
        return cells, ids

    def _fuse_stubborn(self, obs, pix, cells, goal_id, agent, lower):
        return self._fuse_argmax(obs, pix, cells, goal_id, agent, lower)

    def _goal_statistics(self, obs, pix, cells, ids, goal_id):
        """mapper.py:448-515: four running statistics per cell; collection of feature rows, or the
        naive-Bayes veto of goal labels."""
        self.total_view[cells] += 1
        goal_classes = self.objnav_to_sem_pred[goal_id]
        prob = obs["semantic_prob"].transpose(1, 2, 0)
        n_classes = prob.shape[-1]
        others = np.delete(np.arange(n_classes), goal_classes)
        flat = prob.reshape(-1, n_classes)
        goal_mass = np.sum(flat[:, goal_classes], axis=-1)
        self.cumulative_conf[cells] += goal_mass[pix]
        self.max_conf[cells] = np.maximum(self.max_conf[cells], goal_mass[pix])
        other_max = np.max(flat[:, others], axis=-1)
        self.max_other_conf[cells] = np.maximum(self.max_other_conf[cells], other_max[pix])
        cols = [self.total_view[cells], self.cumulative_conf[cells], self.max_conf[cells],
                self.max_other_conf[cells]]
        if self.collect_nb_data:
            is_goal = self.gt_map[cells] == goal_id
            if is_goal.any():
                self.eps_nb_data["goal"].append(np.stack([c[is_goal] for c in cols], axis=-1))
            if not is_goal.any():
                self.eps_nb_data["non_goal"].append(np.stack([c[~is_goal] for c in cols], axis=-1))
            self.eps_nb_data["goal_id"] = goal_id
        else:
            features = np.stack(cols, axis=-1)
            if len(features) > 0:
                pred = gaussian_nb_predict(self.nb_models[goal_id], features)
                ids[(pred == 0) & (ids == goal_id)] = 2

    def _fuse_goal_decay(self, obs, pix, cells, goal_id, agent, lower):
        ids = obs["semantic"].reshape(-1)[pix]
        if self.filter_goal_prob is not None:
            self._filter_goal(ids, 1 - obs["semantic_unc"].reshape(-1)[pix], goal_id)
        cells, ids = self._despeckle_goal(cells, ids, goal_id)
        goal = ids == goal_id
        self.goal_map[(cells[0][goal], cells[1][goal])] += 1
        self.goal_map[(cells[0][~goal], cells[1][~goal])] *= self.goal_decay
        show = self.goal_map[cells] > self.goal_mark
        ids[goal] = 0
        ids[show] = goal_id
        self._occupancy_override(ids, cells)
        return cells, ids

    def _despeckle_goal(self, cells, ids, goal_id):
        """mapper.py:1114-1161: erode (ceil(0.1/p2m)) then dilate (ceil(0.2/p2m)) the goal
        mask on the bounding box of this frame's cells; the update set becomes the
        row-major list of {dilated} | {non-goal, non-zero} cells of that box."""
        if not np.any(ids == goal_id):
            return cells, ids
        r0, r1 = cells[0].min(), cells[0].max()
        c0, c1 = cells[1].min(), cells[1].max()
        local = np.zeros((r1 - r0 + 1, c1 - c0 + 1), dtype=np.uint32)
        local[cells[0] - r0, cells[1] - c0] = ids
        goal = local == goal_id
        eroded = raster.erode_rect(goal, int(math.ceil(0.1 / self.pixel2meter)))
        grown = raster.dilate_same(eroded, int(math.ceil(0.2 / self.pixel2meter)))
        keep = grown | (local != goal_id) & (local != 0)
        rc = np.argwhere(keep)
        return (rc[:, 0] + r0, rc[:, 1] + c0), local[keep]

    def _fuse_avg(self, obs, pix, cells, weight_kind):
        c = obs["semantic_prob"].shape[0]
        p = obs["semantic_prob"].transpose(1, 2, 0).reshape(-1, c)[pix]
        if weight_kind == "w":
            w = 1 - scipy.stats.entropy(p, axis=-1, base=c)[..., np.newaxis]
        elif weight_kind == "w2":
            w = 1 / scipy.stats.entropy(p, axis=-1, base=c)[..., np.newaxis]
        else:
            w = 1
        self.avg_prob_weight_map[cells] += w
        self.avg_prob_seg_map[cells] += (
            (1 / self.avg_prob_weight_map[cells]) * (p - self.avg_prob_seg_map[cells]) * w)
        fused = self.avg_prob_seg_map[cells]
        self.unc_map[cells] = scipy.stats.entropy(fused, axis=-1, base=c)[..., np.newaxis]
        ids = self._schema(np.argmax(fused, axis=-1))
        self._occupancy_override(ids, cells)
        return cells, ids

    def _fuse_avg_prob(self, obs, pix, cells, goal_id, agent, lower):
        return self._fuse_avg(obs, pix, cells, "")

    def _fuse_w_avg_prob(self, obs, pix, cells, goal_id, agent, lower):
        return self._fuse_avg(obs, pix, cells, "w")

    def _fuse_w2_avg_prob(self, obs, pix, cells, goal_id, agent, lower):
        return self._fuse_avg(obs, pix, cells, "w2")

    def _fuse_log_odds(self, obs, pix, cells, goal_id, agent, lower):
        c = obs["semantic_prob"].shape[0]
        p = obs["semantic_prob"].transpose(1, 2, 0).reshape(-1, c)[pix]
        self.log_odds_seg_map[cells] += np.log(p / p[..., 0:1])
        ids = self._schema(np.argmax(self.log_odds_seg_map[cells], axis=-1))
        self._occupancy_override(ids, cells)
        return cells, ids

    def _fuse_hits_views(self, obs, pix, cells, goal_id, agent, lower, weighted):
        sem = obs["semantic"].reshape(-1)
        ids = sem[pix]
        if weighted:
            self.unc_map[cells] = obs["semantic_unc"].reshape(-1, 1)[pix]
        elif self.filter_goal_prob is not None:
            self._filter_goal(ids, obs["semantic_max_prob"].reshape(-1)[pix], goal_id)
        self._occupancy_override(ids, cells)
        bg = ids != goal_id
        self.background_map[(cells[0][bg], cells[1][bg])] = ids[bg]

        gated, gpix = cells, pix
        if self.hv_limit_obj_view_dist > 0.0:
            xz = np.stack([cells[1], cells[0]], axis=-1) * self.pixel2meter + lower[[0, 2]]
            near = np.linalg.norm(xz - np.array([agent[0], agent[2]]), axis=-1) < self.hv_limit_obj_view_dist
            gated, gpix = (cells[0][near], cells[1][near]), pix[near]
        is_goal = sem[gpix] == goal_id
        goal_cells = (gated[0][is_goal], gated[1][is_goal])
        if weighted:
            conf = obs["semantic_max_prob"].reshape(-1)
            self.views[gated] += conf[gpix]
            self.hits[goal_cells] += conf[gpix[is_goal]]
        else:
            self.views[cells] += 1
            self.hits[goal_cells] += 1
        show = (self.hits[gated] > 0) & (self.views[gated] != np.inf)
        self.semantic_map = self.background_map.copy()
        self.semantic_map[(gated[0][show], gated[1][show])] = goal_id
        return cells, ids

    def _fuse_hv(self, obs, pix, cells, goal_id, agent, lower):
        return self._fuse_hits_views(obs, pix, cells, goal_id, agent, lower, False)

    def _fuse_prob_hv(self, obs, pix, cells, goal_id, agent, lower):
        return self._fuse_hits_views(obs, pix, cells, goal_id, agent, lower, True)

    # ---- stage 5 (a-18)
    def _found(self, agent, lower, goal_id, hv_rule):
        robot = cell_of(agent[:3], lower, self.pixel2meter)
        radius = int(self.found_dist_m / self.pixel2meter)
        disk = raster.disk_mask(self.semantic_map.shape, int(robot[0]), int(robot[1]), radius)
        self.last_disk = disk
        if hv_rule:
            cand = tuple(np.argwhere(disk & (self.views > self.hv_min_views) & (self.hits > 0)).T)
            v, h = self.views[cand], self.hits[cand]
            live = v != np.inf
            with np.errstate(divide="ignore", invalid="ignore"):
                strong = (h / v) > self.hv_ratio
            if np.any(live):
                rej = live & ~strong
                rr, cc = cand[0][rej], cand[1][rej]
                self.views[rr, cc] = np.inf
                self.semantic_map[rr, cc] = self.background_map[rr, cc]
                self.semantic_map_color[rr, cc] = self.color_pallete[self.background_map[rr, cc]]
            return True if np.any(live & strong) else None
        in_disk = tuple(np.argwhere(disk).T)
        goal = self.semantic_map[in_disk] == goal_id
        if self.probablistic_mapping and self.rule in AVG_RULES:
            c = self.avg_prob_seg_map.shape[-1]
            ent = scipy.stats.entropy(self.avg_prob_seg_map[in_disk][goal], axis=-1, base=c)
            return True if np.any(ent < self.entropy_threshold) else None
        if self.probablistic_mapping and self.rule == "log_odds":
            c = self.log_odds_seg_map.shape[-1]
            pr = scipy.special.softmax(self.log_odds_seg_map[in_disk][goal], axis=-1)
            ent = scipy.stats.entropy(pr, axis=-1, base=c)
            return True if np.any(ent < self.entropy_threshold) else None
        return True if np.any(goal) else None


def gaussian_nb_predict(model, features):
    """sklearn GaussianNB.predict restated (naive_bayes.py, _joint_log_likelihood): per class
    log prior - 0.5 sum log(2 pi var) - 0.5 sum (x - theta)^2 / var, label of the first largest."""
    get = (lambda k: model[k]) if isinstance(model, dict) else (lambda k: getattr(model, k))
    classes, prior = np.asarray(get("classes_")), np.asarray(get("class_prior_"), dtype=np.float64)
    theta, var = np.asarray(get("theta_"), dtype=np.float64), np.asarray(get("var_"), dtype=np.float64)
    x = np.asarray(features, dtype=np.float64)
    jll = []
    for i in range(classes.shape[0]):
        jointi = np.log(prior[i])
        n_ij = -0.5 * np.sum(np.log(2.0 * np.pi * var[i, :]))
        n_ij = n_ij - 0.5 * np.sum(((x - theta[i, :]) ** 2) / (var[i, :]), 1)
        jll.append(jointi + n_ij)
    jll = np.array(jll).T
    return classes[np.argmax(jll, axis=1)]


def palette(n):
    """PASCAL-VOC style bit-interleaved colour table (mapper.py:1217-1231)."""
    table = np.zeros((n, 3), dtype=np.uint8)
    for i in range(n):
        v, rgb = i, [0, 0, 0]
        for j in range(8):
            for ch in range(3):
                rgb[ch] |= ((v >> ch) & 1) << (7 - j)
            v >>= 3
        table[i] = rgb
    return table
