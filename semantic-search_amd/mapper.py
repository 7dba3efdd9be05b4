"""`SemanticMapper`: the reference's class interface over the HIP engine (one environment).

Same constructor keywords, methods, late-mutable attributes and public arrays as
sem_objnav/obj_nav/mapper.py:48-269, so the callers (`MappingObjNavEnv`, env.py:165-185,
377, 403, 560; `eval_objnav`, eval_objnav.py:118-137) can construct it in place of the
reference class.  Observations arrive as numpy arrays (one frame), are copied to the GPU,
fused there, and the map arrays are copied back only when a caller reads them.

The "stubborn" rule takes its naive-Bayes models as fitted arrays (`nb_models`, or a JSON file
written by tools/export_nb_models.py) instead of the reference's pickle (mapper.py:184-191).
The path overlay (mapper.py:822-833) draws its line with cv2.line when OpenCV is importable and
with an approximate thick line otherwise (`_line_cells`).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import os

import numpy as np
import torch

from . import engine as _engine

_TORCH_DTYPES = {np.float32: torch.float32, np.int32: torch.int32}

_MAP_NAMES = ("semantic_map", "semantic_map_color", "height_map", "gt_map", "unc_map", "goal_map",
              "log_odds_seg_map", "avg_prob_seg_map", "avg_prob_weight_map", "background_map",
              "hits", "views", "total_view", "cumulative_conf", "max_conf", "max_other_conf")
_RULE_ATTRS = ("entropy_threshold", "goal_decay", "goal_mark", "hv_ratio", "hv_min_views",
               "hv_limit_obj_view_dist")


def _box_cells(center, sizes, lower, pixel2meter, shape):
    """Half-open cell rectangle of an axis-aligned box: both corners through world2map, every
    index clipped to the map (mapper.py:1063-1072)."""
    lo = np.asarray(center, dtype=np.float64) - np.asarray(sizes, dtype=np.float64) - lower
    hi = np.asarray(center, dtype=np.float64) + np.asarray(sizes, dtype=np.float64) - lower
    r1, c1 = int(np.floor(lo[2] / pixel2meter)), int(np.floor(lo[0] / pixel2meter))
    r2, c2 = int(np.floor(hi[2] / pixel2meter)), int(np.floor(hi[0] / pixel2meter))
    clip = lambda v, n: min(max(v, 0), n - 1)
    return clip(r1, shape[0]), clip(r2, shape[0]), clip(c1, shape[1]), clip(c2, shape[1])


def _dilate_box(mask, size):
    """Binary dilation by a size x size box anchored at its centre, nothing outside the map."""
    h, w = mask.shape
    a = size // 2
    pad = np.zeros((h + size, w + size), dtype=bool)
    pad[a:a + h, a:a + w] = mask
    out = np.zeros((h, w), dtype=bool)
    for i in range(size):
        for j in range(size):
            out |= pad[i:i + h, j:j + w]
    return out


def fp_fn_detections(semantic_map, gt_map, goal_sem_id, centers, sizes, lower, pixel2meter,
                     hits=None, views=None):
    """(fp, fn, is_fn_bbox) of mapper.py:1053-1112 on host arrays.  A box counts only if the
    ground-truth map holds the goal inside it; it is a false negative if the map shows no goal
    there (hit/view rules: no cell with hits > 0 and finite views).  False positives are the
    8-connected blobs of predicted goal cells outside all counted boxes, after a box dilation of
    int(0.1 / pixel2meter) cells."""
    import scipy.ndimage
    lower = np.asarray(lower, dtype=np.float64)
    shape = semantic_map.shape
    pred = semantic_map == goal_sem_id
    counted = np.zeros(shape, dtype=bool)
    fn, is_fn = 0, []
    for center, size in zip(centers, sizes):
        r1, r2, c1, c2 = _box_cells(center, size, lower, pixel2meter, shape)
        if not (gt_map[r1:r2, c1:c2] == goal_sem_id).any():
            is_fn.append(False)
            continue
        counted[r1:r2, c1:c2] = True
        if hits is not None:
            seen = bool(((hits[r1:r2, c1:c2] > 0) & (views[r1:r2, c1:c2] != np.inf)).any())
        else:
            seen = bool(pred[r1:r2, c1:c2].any())
        is_fn.append(not seen)
        fn += 0 if seen else 1
    fp_mask = _dilate_box(pred & ~counted, int(0.1 / pixel2meter))
    _, fp = scipy.ndimage.label(fp_mask, scipy.ndimage.generate_binary_structure(2, 2))
    return int(fp), fn, is_fn


def _draw_line(mask, p0, p1, color, thickness):
    """cv2.line(mask, p0, p1, color, thickness) (mapper.py:823-830); points are (x, y).  Without
    OpenCV: the 8-connected centre line of OpenCV's LineIterator under a square brush of
    `thickness` cells, an approximation of the rectangle-plus-round-caps OpenCV fills."""
    try:
        import cv2
    except ImportError:
        return _line_cells(mask, p0, p1, color, thickness)
    return cv2.line(mask, (int(p0[0]), int(p0[1])), (int(p1[0]), int(p1[1])), color, thickness)


def _line_cells(mask, p0, p1, color, thickness):
    x0, y0, x1, y1 = int(p0[0]), int(p0[1]), int(p1[0]), int(p1[1])
    if x1 < x0:
        x0, y0, x1, y1 = x1, y1, x0, y0
    dx, dy = x1 - x0, abs(y1 - y0)
    ystep = -1 if y1 < y0 else 1
    steep = dy > dx
    major, minor = (dy, dx) if steep else (dx, dy)
    err, plus, minus = major - 2 * minor, 2 * major, -2 * minor
    x, y = x0, y0
    lo, hi = -((max(thickness, 1) - 1) // 2), max(thickness, 1) // 2
    h, w = mask.shape[:2]
    for _ in range(major + 1):
        r0, r1, c0, c1 = max(y + lo, 0), min(y + hi, h - 1), max(x + lo, 0), min(x + hi, w - 1)
        if r0 <= r1 and c0 <= c1:
            mask[r0:r1 + 1, c0:c1 + 1] = color
        bump = err < 0
        err += minus + (plus if bump else 0)
        if steep:
            y += ystep
            x += 1 if bump else 0
        else:
            x += 1
            y += ystep if bump else 0
    return mask


def _load_nb_models(nb_models, model_name):
    """{goal id: model} from a dict of fitted models or from a JSON file
    {"<goal id>": {"classes_": [...], "class_prior_": [...], "theta_": [[...]], "var_": [[...]]}}."""
    import json
    import os
    if nb_models is None and model_name is not None:
        nb_models = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nb_models", f"{model_name}.json")
    if nb_models is None:
        raise ValueError("the stubborn rule needs nb_models (fitted GaussianNB per goal id) or model_name")
    if isinstance(nb_models, (str, os.PathLike)):
        if not os.path.exists(nb_models):
            raise FileNotFoundError(
                f"{nb_models}: export the reference's naive-Bayes models to JSON first "
                "(tools/export_nb_models.py, run where the pickle can be trusted)")
        with open(nb_models) as fh:
            nb_models = {int(k): v for k, v in json.load(fh).items()}
    return dict(nb_models)


def sp_found_override(true_found, in_disk, anywhere, distance_to_vp, success_distance, fp_found):
    """The shortest-path policy's override of the found decision (env.py:574-605), element-wise
    over envs.  `true_found` = the mapper's decision, `in_disk` / `anywhere` = the goal id lies in
    gt_map inside the found disk / at all, `fp_found` = the episode's false-positive latch so far.
    Returns (is_found, fp_found, fn_found): fn_found is None where the robot is not within
    `success_distance` of the viewpoint (the reference leaves its flag untouched there)."""
    true_found = np.asarray(true_found, dtype=bool)
    in_disk = np.asarray(in_disk, dtype=bool)
    anywhere = np.asarray(anywhere, dtype=bool)
    fp = np.asarray(fp_found, dtype=bool) | (true_found & ~in_disk)          # env.py:580-594
    near = np.asarray(distance_to_vp, dtype=np.float64) < success_distance   # env.py:598
    fn = np.where(near, anywhere & ~true_found & ~fp, None)                  # env.py:599-603
    return near & anywhere, fp, fn                                           # env.py:575, 604-605


class SpPolicyFoundOverride:
    """State and call of that override for one SemanticMapper, as MappingObjNavEnv keeps it
    (`_sp_fp_found`, `_sp_fn_found`, reset with the episode: env.py:317-318, 564-605; they end up as the
    episode's `fp_found` / `fn_found`, env.py:724-734)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.sp_fp_found = False
        self.sp_fn_found = False

    def __call__(self, mapper, true_found, distance_to_vp, success_distance) -> bool:
        """After `mapper.update_map(...)` returned `true_found`: the overridden decision.  The two
        map questions are answered on the GPU from the maps in place (only those the branch reads)."""
        near = distance_to_vp < success_distance
        # the engine answers for all of its envs: this mapper's own is env `mapper._e` (0 for a
        # mapper of its own, the view's index inside a MapperGroup)
        e = mapper._e
        in_disk, anywhere = mapper._engine.gt_goal_query(in_disk=bool(true_found), anywhere=bool(near))
        found, fp, fn = sp_found_override([bool(true_found)], [True] if in_disk is None else in_disk[e:e + 1],
                                          [False] if anywhere is None else anywhere[e:e + 1], [distance_to_vp],
                                          success_distance, [self.sp_fp_found])
        self.sp_fp_found = bool(fp[0])
        if fn[0] is not None:
            self.sp_fn_found = bool(fn[0])
        return bool(found[0])


class SemanticMapper:
    _e = 0   # this mapper's env inside its engine (a view of a MapperGroup has its own)

    def __init__(
        self,
        num_semantic_classes: int,
        depth_hfov: float,
        depth_range: Tuple[float, float],
        egomap_small_side_m: float = 3.0,
        egomap_large_side_m: float = 10.0,
        egomap_size: Tuple[int, int] = (128, 128),
        pixel2meter: float = 0.03,
        aux_num_bins: int = 12,
        aux_radius_m: float = 0.4,
        aux_bin_m: float = 0.15,
        path_width_m: float = 0.1,
        height_cutoff_m: float = 0.1,
        agent_draw_radius_m: float = 0.1,
        entropy_threshold: float = 0.5,
        probablistic_mapping: bool = False,
        prob_aggregate: str = "height_map_argmax",
        success_dist_m: float = 1.0,
        seg_model_n_classes: Optional[int] = None,
        filter_goal_prob: Optional[float] = None,
        collect_nb_data: bool = False,
        model_name: Optional[str] = None,
        device: str = "cuda:0",
        nb_models=None,
    ):
        self.pixel2meter = pixel2meter
        self.depth_range = depth_range
        self.depth_hfov = depth_hfov
        self.aux_bin_number = aux_num_bins
        self.bin_size = 360 / aux_num_bins
        self.angle_to_bins = np.arange(self.bin_size, 360 + self.bin_size, self.bin_size)
        self.egomap_size = egomap_size
        self.egomap_small_crop = [int(egomap_small_side_m / pixel2meter)] * 2
        self.egomap_large_crop = [int(egomap_large_side_m / pixel2meter)] * 2
        self.offset_for_cut = 150
        self.aux_bin_px = int(aux_bin_m / pixel2meter)
        self._ego_cfg = dict(egomap_small_side_m=egomap_small_side_m, egomap_large_side_m=egomap_large_side_m,
                             agent_draw_radius_m=agent_draw_radius_m, aux_radius_m=aux_radius_m, aux_bin_m=aux_bin_m)
        self.path_width_px = int(path_width_m / pixel2meter)
        self.agent_radius_px = int(agent_draw_radius_m / pixel2meter)
        # offsets of the auxiliary-angle squares from the image centre, in pixel coordinates
        # (mapper.py:82-89: pol2cart of the bin angle, negated, truncated by the int32 cast)
        aux_radius_px = int(aux_radius_m / pixel2meter)
        self.aux_points = np.array(
            [-np.array((aux_radius_px * np.cos(np.deg2rad(deg)), aux_radius_px * np.sin(np.deg2rad(deg))))
             for deg in np.arange(0, 360, self.bin_size)], dtype=np.int32)
        self.color_pallete = _engine.colormap(num_semantic_classes + 3)
        self.active_goal_color = self.color_pallete[num_semantic_classes]          # mapper.py:103
        self.path_color = np.array([128, 0, 128], dtype=np.uint8)
        self.color_pallete[num_semantic_classes + 1] = self.path_color
        self.arrow_color = tuple(int(c) for c in self.color_pallete[num_semantic_classes + 2])   # mapper.py:106-108
        self.aux_pred_color = np.array([255, 255, 255], dtype=np.uint8)             # mapper.py:109
        self.log_data = {}                                                          # mapper.py:144
        self.path_id = num_semantic_classes + 1
        self.height_cutoff_m = height_cutoff_m
        self.probablistic_mapping = probablistic_mapping
        self.num_semantic_classes = num_semantic_classes
        self.prob_aggregate = prob_aggregate
        self.found_dist_m = success_dist_m
        self.nb_models = None
        if prob_aggregate in ("height_map_stubborn", "stubborn"):
            # the reference unpickles nb_models/<model_name>.pkl here (mapper.py:184-191); this class
            # takes the fitted models themselves: {goal id: GaussianNB-like}, or the path of a JSON
            # file holding classes_, class_prior_, theta_, var_ per goal id
            self.nb_models = _load_nb_models(nb_models, model_name)
        self._rule, self._use_height = _engine.split_rule(prob_aggregate, probablistic_mapping)
        # rule constants exist only for the rule that uses them, as in the reference
        # (mapper.py:145-169); callers test them with hasattr (eval_objnav.py:118-129)
        if probablistic_mapping and self._rule in ("hv", "prob_hv"):
            self.hv_limit_obj_view_dist = 0.0
            self.hv_min_views = 10
            self.hv_ratio = 0.5
        elif probablistic_mapping and self._rule in _engine.PROB_RULES:
            self.entropy_threshold = entropy_threshold
        elif self._rule == "goal_decay":
            self.goal_decay = 0.9
            self.goal_mark = 2.0
        if num_semantic_classes == 9:
            self.goal2sem_id = np.array([3, 4, 5, 6, 7, 8])
        elif num_semantic_classes in (41, 38):
            self.goal2sem_id = np.array([3, 11, 14, 18, 22, 10])
        self.map_seg_to_schema = None
        self.objnav_to_sem_pred = None
        self.sem_model_n_classes = seg_model_n_classes
        self.filter_goal_prob = filter_goal_prob
        self.collect_nb_data = collect_nb_data
        self.found_idx = None
        self.prev_pos_px = None
        self.map_bounds = None
        self._device = device
        self._engine: Optional[_engine.MapEngine] = None
        self._engine_key = None
        self._host_cache: Dict[str, np.ndarray] = {}
        self._stage: Dict[str, torch.Tensor] = {}
        self.pinned_staging = os.environ.get("SSMAP_PINNED_STAGING", "1") != "0"
        self.staging_chunks = int(os.environ.get("SSMAP_STAGING_CHUNKS", "4"))
        self._bound = False
        self._schema_sent = False
        self.last_status = 0
        self.last_n_updates = 0

    # ------------------------------------------------------------------ engine plumbing
    def _ensure_engine(self, h: int, w: int):
        rows, cols = _engine.map_size(self.map_bounds, self.pixel2meter)
        key = (h, w)
        eng = self._engine
        if eng is None or self._engine_key != key or rows > eng.max_map_hw[0] or cols > eng.max_map_hw[1]:
            if eng is not None:
                eng.close()
            eng = _engine.MapEngine(
                1, (h, w), num_semantic_classes=self.num_semantic_classes, depth_hfov=self.depth_hfov,
                depth_range=self.depth_range, pixel2meter=self.pixel2meter,
                height_cutoff_m=self.height_cutoff_m, probablistic_mapping=self.probablistic_mapping,
                prob_aggregate=self.prob_aggregate, seg_model_n_classes=self.sem_model_n_classes,
                max_map_hw=(rows, cols), success_dist_m=self.found_dist_m,
                collect_nb_data=self.collect_nb_data, device=self._device)
            for goal, model in (self.nb_models or {}).items():
                eng.set_nb_model(int(goal), model)
            eng.set_egomap(self._ego_cfg["egomap_small_side_m"], self._ego_cfg["egomap_large_side_m"],
                           self.egomap_size, self._ego_cfg["agent_draw_radius_m"], self.aux_bin_number,
                           self._ego_cfg["aux_radius_m"], self._ego_cfg["aux_bin_m"], self.offset_for_cut)
            self._engine, self._engine_key = eng, key
            self._bound = False
            self._schema_sent = False
        if not self._bound:
            eng.reset_env(0, self.map_bounds)
            self._bound = True
        if not self._schema_sent and self.sem_model_n_classes and self.map_seg_to_schema is not None:
            # map_seg_to_schema is a callable on label arrays (env.py:183); probe it once
            lut = np.asarray(self.map_seg_to_schema(np.arange(self.sem_model_n_classes)))
            pools = None
            if self.objnav_to_sem_pred is not None:
                pools = {g: list(self.objnav_to_sem_pred[g]) for g in range(3, 9)
                         if g in self.objnav_to_sem_pred}
            eng.set_schema(lut, pools)
            self._schema_sent = True
        return eng

    def _sync_params(self, eng):
        p = eng.params
        for name in _RULE_ATTRS + ("filter_goal_prob", "found_dist_m"):
            if hasattr(self, name) and getattr(p, name) != getattr(self, name):
                setattr(p, name, getattr(self, name))

    # ------------------------------------------------------------------ reference interface
    def reset(self, map_bounds: np.ndarray):
        if self.collect_nb_data:
            self.eps_nb_data = {"goal": [], "non_goal": [], "goal_id": None}   # mapper.py:265
        self.map_bounds = np.asarray(map_bounds)
        self._bound = False
        self._host_cache = {}
        self.prev_pos_px = None
        self.found_idx = None
        self.path_mask = np.zeros(_engine.map_size(self.map_bounds, self.pixel2meter), dtype=np.uint8)
        if self._engine is not None:
            rows, cols = _engine.map_size(self.map_bounds, self.pixel2meter)
            if rows <= self._engine.max_map_hw[0] and cols <= self._engine.max_map_hw[1]:
                self._engine.reset_env(0, self.map_bounds)
                self._bound = True

    def _inputs(self, obs, dev):
        """One frame's arrays as device tensors (no batch axis) under the engine's keyword names,
        and the goal's schema id.  numpy arrays like the reference's callers pass, or torch tensors
        (a segmentation model's outputs that are already on the GPU skip the round trip through the
        host)."""
        goal_id = int(self.goal2sem_id[obs["objectgoal"]][0])

        def up(key, dtype):
            if key not in obs or obs[key] is None:
                return None
            val = obs[key]
            if isinstance(val, torch.Tensor):
                return val.detach().to(device=dev, dtype=_TORCH_DTYPES[dtype]).contiguous()
            arr = np.ascontiguousarray(val, dtype=dtype)
            if self.pinned_staging and arr.nbytes >= (1 << 20):
                # large host arrays (the class probabilities are 49 MB a frame at 40 classes) go through a
                # page-locked buffer that is kept between frames: a threaded host copy, then one DMA at
                # PCIe rate, instead of the driver's chunked copy out of pageable memory.  The buffer is
                # free again when update_map returns (the step's results are read behind a stream sync).
                buf = self._stage.get(key)
                if buf is None or tuple(buf.shape) != arr.shape or buf.dtype != _TORCH_DTYPES[dtype]:
                    buf = torch.empty(arr.shape, dtype=_TORCH_DTYPES[dtype], pin_memory=True)
                    self._stage[key] = buf
                src = torch.from_numpy(arr)
                if arr.nbytes < (8 << 20) or self.staging_chunks <= 1:
                    buf.copy_(src)
                    return buf.to(dev, non_blocking=True)
                # the largest arrays in pieces: the DMA of one piece runs while the host copies the next
                out = torch.empty(arr.shape, dtype=_TORCH_DTYPES[dtype], device=dev)
                fb, fs, fo = buf.view(-1), src.view(-1), out.view(-1)
                n = fb.numel()
                step = -(-n // self.staging_chunks)
                for lo in range(0, n, step):
                    hi = min(lo + step, n)
                    fb[lo:hi].copy_(fs[lo:hi])
                    fo[lo:hi].copy_(fb[lo:hi], non_blocking=True)
                return out
            return torch.from_numpy(arr).to(dev)

        tens = {"depth": up("depth", np.float32)}
        if self.probablistic_mapping:
            tens["semantic_gt"] = up("semantic_gt", np.int32)
            if self._rule in _engine.PROB_RULES:
                tens["probs"] = up("semantic_prob", np.float32)
            else:
                tens["semantic"] = up("semantic", np.int32)
                tens["semantic_unc"] = up("semantic_unc", np.float32)
                tens["semantic_max_prob"] = up("semantic_max_prob", np.float32)
                if self._rule == "stubborn" or self.collect_nb_data:
                    tens["probs"] = up("semantic_prob", np.float32)
        else:
            tens["semantic"] = up("semantic", np.int32)
        return tens, goal_id

    def _after_step(self, eng, obs, d_depth, found, status, n_upd, goal_id):
        """Host side of one frame after the engine stepped env `self._e`: depth written back, status,
        naive-Bayes rows, found cells, the new cells of the trail.  Returns (is_found, whether the
        trail has to be painted)."""
        depth = obs["depth"]
        # the reference zeroes the sentinels in the caller's array (mapper.py:282-283)
        if isinstance(depth, torch.Tensor):
            if depth.data_ptr() != d_depth.data_ptr():
                depth.copy_(d_depth)
        else:
            depth[...] = d_depth.cpu().numpy()
        self._host_cache = {}
        self.last_status, self.last_n_updates = int(status), int(n_upd)
        if status & 1:
            raise IndexError("index out of bounds for the semantic map (cell index below -dim)")
        if self.collect_nb_data and self.probablistic_mapping and self._rule in ("argmax", "stubborn"):
            self._collect_nb_rows(eng, goal_id)
        lower = self.map_bounds[:3]
        robot_pos_px = self.world2map(np.asarray(obs["agent_position_and_rotation"])[:3], lower)
        is_found = True if found else None
        if is_found:
            self.found_idx = self._disk_indices(robot_pos_px)
        # path overlay, after the found decision (mapper.py:822-839)
        paint = False
        if self.prev_pos_px is not None:
            before = self.path_mask.astype(bool)
            self.path_mask = _draw_line(self.path_mask, (self.prev_pos_px[1], self.prev_pos_px[0]),
                                        (robot_pos_px[1], robot_pos_px[0]), 1, self.path_width_px)
            new = np.flatnonzero(self.path_mask.astype(bool) & ~before)
            if new.size:
                eng.path_append(self._e, new)
            paint = True
        self.prev_pos_px = robot_pos_px
        return is_found, paint

    def update_map(self, obs: Dict[str, np.ndarray], auxillary_angle=None):
        h, w = obs["depth"].shape
        eng = self._ensure_engine(h, w)
        self._sync_params(eng)
        tens, goal_id = self._inputs(obs, eng.device)
        kwargs = {k: (None if t is None else t.unsqueeze(0).contiguous()) for k, t in tens.items()}
        d_depth = kwargs.pop("depth")
        eng.step(d_depth, obs["agent_position_and_rotation"], obs["camera_position_and_rotation"],
                 [goal_id], **kwargs)
        found, status, n_upd = eng.results()
        is_found, paint = self._after_step(eng, obs, d_depth[0], found[0], status[0], n_upd[0], goal_id)
        if paint:
            eng.path_paint()
        small, large, unc_s, unc_l = eng.ego_maps(np.asarray(obs["agent_position_and_rotation"])[None], [goal_id],
                                                  None if auxillary_angle is None else np.asarray(auxillary_angle)[None])
        ego = (small[0].cpu().numpy(), large[0].cpu().numpy(),
               None if unc_s is None else unc_s[0].cpu().numpy(), None if unc_l is None else unc_l[0].cpu().numpy())
        return ego, is_found

    def _collect_nb_rows(self, eng, goal_id):
        """Feature rows of this frame's update cells for fitting the naive-Bayes models
        (mapper.py:470-500): one [k, 4] float64 array of the cells whose ground truth is the goal;
        the other cells' rows only when no update cell is a goal cell."""
        pix, cells = eng.update_list(self._e, self.last_n_updates)     # ascending pixel order, like valid_flat_idx
        maps = eng.maps[self._e]
        idx = torch.from_numpy(cells).to(eng.device)
        cols = [maps[name].reshape(-1)[idx].cpu().numpy() for name in
                ("total_view", "cumulative_conf", "max_conf", "max_other_conf")]
        cols[0] = cols[0].view(np.uint32)
        is_goal = maps["gt_map"].reshape(-1)[idx].cpu().numpy().view(np.uint32) == goal_id
        if is_goal.any():
            self.eps_nb_data["goal"].append(np.stack([c[is_goal] for c in cols], axis=-1))
        if not is_goal.any():
            self.eps_nb_data["non_goal"].append(np.stack([c[~is_goal] for c in cols], axis=-1))
        self.eps_nb_data["goal_id"] = goal_id

    def _disk_indices(self, center):
        """Row/column index arrays of the found disk (mapper.py:737-743, 785)."""
        radius = int(self.found_dist_m / self.pixel2meter)
        rows, cols = _engine.map_size(self.map_bounds, self.pixel2meter)
        hw = np.full(radius + 1, -1, dtype=np.int64)
        err, dx, dy, plus, minus = 0, radius, 0, 1, 2 * radius - 1
        while dx >= dy:
            hw[dy] = max(hw[dy], dx)
            hw[dx] = max(hw[dx], dy)
            dy += 1
            err += plus
            plus += 2
            if err > 0:
                err -= minus
                dx -= 1
                minus -= 2
        rr, cc = [], []
        for k in range(-radius, radius + 1):
            r = int(center[0]) + k
            if r < 0 or r >= rows or hw[abs(k)] < 0:
                continue
            c0, c1 = max(int(center[1]) - hw[abs(k)], 0), min(int(center[1]) + hw[abs(k)], cols - 1)
            if c0 <= c1:
                rr.append(np.full(c1 - c0 + 1, r))
                cc.append(np.arange(c0, c1 + 1))
        if not rr:
            return (np.zeros(0, np.int64), np.zeros(0, np.int64))
        return (np.concatenate(rr), np.concatenate(cc))

    def world2map(self, world_coords_m: np.ndarray, world_lower_bound_m: np.ndarray) -> np.ndarray:
        d = world_coords_m - world_lower_bound_m
        uv = np.stack([d[..., 2], d[..., 0]], axis=-1)
        return np.floor(uv / self.pixel2meter).astype(np.int64)

    def map2world(self, map_coords: np.ndarray, world_lower_bound_m: np.ndarray) -> np.ndarray:
        uv = np.stack([map_coords[..., 1], map_coords[..., 0]], axis=-1) * self.pixel2meter
        return uv + world_lower_bound_m[[0, 2]]

    def compute_fp_fn_detections(self, task_goal_ids, bboxes):
        """Per-episode detection metric (mapper.py:1053-1112): false-negative goal boxes and
        the number of false-positive goal blobs on the map.  `bboxes` have `.center` and
        `.sizes` (world metres).  Runs on the host over the downloaded maps, once per episode."""
        goal_sem_id = self.goal2sem_id[task_goal_ids][0]
        hv = self.probablistic_mapping and ("hv" in self.prob_aggregate)
        semantic_map = self.semantic_map
        gt_map = self.gt_map if self.probablistic_mapping else semantic_map   # alias, mapper.py:257
        return fp_fn_detections(semantic_map, gt_map, goal_sem_id, [b.center for b in bboxes],
                                [b.sizes for b in bboxes], self.map_bounds[:3], self.pixel2meter,
                                hits=self.hits if hv else None, views=self.views if hv else None)

    # ------------------------------------------------------------------ public map arrays
    def _host(self, name):
        if self._engine is None or not getattr(self, "_bound", False):
            return None
        maps = self._engine.maps[self._e]
        if name not in maps:
            return None
        if name not in self._host_cache:
            arr = maps[name].cpu().numpy()
            if arr.dtype == np.int32:
                arr = arr.view(np.uint32)
            self._host_cache[name] = arr
        return self._host_cache[name]

    def __getattr__(self, name):
        # only reached when normal lookup fails: serve the map arrays from the device
        if name in _MAP_NAMES:
            val = self._host(name)
            if val is None and name != "unc_map":
                raise AttributeError(name)
            return val
        raise AttributeError(name)
