"""Batched map-fusion engine: E independent environments stepped by one C-ABI call.

Host-side mirror of the reference's `SemanticMapper` state (sem_objnav/obj_nav/mapper.py):
the same map arrays with the same names, dtypes and layouts (mapper.py:205-255), held as
torch tensors on one MI355X, plus the late-mutable rule attributes
(eval_objnav.py:118-137).  All arithmetic happens in the HIP library (include/ssmap.h).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _capi

AVG_RULES = ("avg_prob", "w_avg_prob", "w2_avg_prob")
PROB_RULES = AVG_RULES + ("log_odds",)


def split_rule(prob_aggregate: str, probablistic_mapping: bool):
    """('height_map_avg_prob', True) -> ('avg_prob', use_height=True)."""
    use_height = prob_aggregate.startswith("height_map_")
    rule = prob_aggregate[len("height_map_"):] if use_height else prob_aggregate
    if not probablistic_mapping:
        # direct mapping applies the height override only under this exact name (mapper.py:346)
        return "direct", prob_aggregate == "height_map_argmax"
    if rule not in _capi.RULES or rule == "direct":
        raise RuntimeError("Invalid prob_aggregate")
    return rule, use_height


def focal_lengths(height: int, width: int, hfov_deg: float):
    """fx, fy exactly as pointcloud() derives them (mapper.py:21-25)."""
    hfov = np.deg2rad(hfov_deg)
    fx = width / (2.0 * np.tan(hfov / 2.0))
    vfov = height / width * hfov
    fy = height / (2.0 * np.tan(vfov / 2.0))
    return float(fx), float(fy)


def colormap(n: int) -> np.ndarray:
    """Bit-interleaved colour table of the reference (mapper.py:1217-1231)."""
    table = np.zeros((n, 3), dtype=np.uint8)
    for i in range(n):
        v = i
        for j in range(8):
            for ch in range(3):
                table[i, ch] |= ((v >> ch) & 1) << (7 - j)
            v >>= 3
    return table


def map_size(map_bounds, pixel2meter):
    """(rows, cols) of reset() (mapper.py:197-204): float64 ceil, rows <-> z, cols <-> x."""
    b = np.asarray(map_bounds)
    rows = int(np.ceil(np.abs(b[5] - b[2]) / pixel2meter).astype(np.int32))
    cols = int(np.ceil(np.abs(b[3] - b[0]) / pixel2meter).astype(np.int32))
    return rows, cols


class RuleParams:
    """Attributes callers overwrite after construction; every write marks them dirty so the
    next step re-sends them."""
    _FIELDS = ("entropy_threshold", "filter_goal_prob", "goal_decay", "goal_mark", "hv_ratio",
               "hv_min_views", "hv_limit_obj_view_dist", "found_dist_m", "collect_nb_data")

    def __init__(self, **kw):
        object.__setattr__(self, "dirty", True)
        for k in self._FIELDS:
            object.__setattr__(self, k, kw.get(k))

    def __setattr__(self, key, value):
        object.__setattr__(self, key, value)
        object.__setattr__(self, "dirty", True)

    def pack(self) -> _capi.Params:
        p = _capi.Params()
        p.entropy_threshold = float(self.entropy_threshold if self.entropy_threshold is not None else 0.5)
        p.use_filter = 0 if self.filter_goal_prob is None else 1
        p.filter_goal_prob = float(self.filter_goal_prob or 0.0)
        p.goal_decay = float(self.goal_decay)
        p.goal_mark = float(self.goal_mark)
        p.hv_ratio = float(self.hv_ratio)
        p.hv_min_views = float(self.hv_min_views)
        p.hv_limit_obj_view_dist = float(self.hv_limit_obj_view_dist)
        p.found_dist_m = float(self.found_dist_m)
        p.nb_stats = 1 if self.collect_nb_data else 0
        return p


class MapEngine:
    """E environments, one frame each per `step`."""

    def __init__(self, n_envs: int, img_hw, *, num_semantic_classes: int = 9, depth_hfov: float,
                 depth_range, pixel2meter: float = 0.03, height_cutoff_m: float = 0.1,
                 probablistic_mapping: bool = False, prob_aggregate: str = "height_map_argmax",
                 seg_model_n_classes: Optional[int] = None, max_map_hw=(480, 480),
                 entropy_threshold: float = 0.5, success_dist_m: float = 1.0,
                 filter_goal_prob: Optional[float] = None, collect_nb_data: bool = False,
                 device="cuda:0"):
        self.lib = _capi.load()
        if not torch.cuda.is_available():
            raise RuntimeError("MapEngine needs a GPU: the fusion path has no CPU fallback")
        self.device = torch.device(device)
        self.n_envs = int(n_envs)
        self.img_h, self.img_w = int(img_hw[0]), int(img_hw[1])
        self.num_semantic_classes = int(num_semantic_classes)
        self.prob_aggregate = prob_aggregate
        self.probablistic_mapping = bool(probablistic_mapping)
        self.rule, self.use_height = split_rule(prob_aggregate, probablistic_mapping)
        self.n_classes = int(seg_model_n_classes or 0)
        self.pixel2meter = float(pixel2meter)
        self.depth_range = depth_range
        self.depth_hfov = depth_hfov
        self.max_map_hw = (int(max_map_hw[0]), int(max_map_hw[1]))
        self.params = RuleParams(entropy_threshold=entropy_threshold, filter_goal_prob=filter_goal_prob,
                                 goal_decay=0.9, goal_mark=2.0, hv_ratio=0.5, hv_min_views=10,
                                 hv_limit_obj_view_dist=0.0, found_dist_m=success_dist_m,
                                 collect_nb_data=bool(collect_nb_data))
        self.collect_nb_data = bool(collect_nb_data)
        fx, fy = focal_lengths(self.img_h, self.img_w, depth_hfov)
        cfg = _capi.Config(
            n_envs=self.n_envs, img_h=self.img_h, img_w=self.img_w, n_classes=self.n_classes,
            n_schema=self.num_semantic_classes, rule=_capi.RULES[self.rule],
            use_height_map=int(self.use_height), max_map_h=self.max_map_hw[0],
            max_map_w=self.max_map_hw[1], depth_min=float(np.float32(depth_range[0])),
            depth_max=float(np.float32(depth_range[1])), fx=fx, fy=fy,
            pixel2meter=self.pixel2meter, height_cutoff_m=float(height_cutoff_m))
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _capi.check(self.lib.ssmap_create(C.byref(cfg), C.byref(self._h)))
            self.color_pallete = colormap(self.num_semantic_classes + 3)
            self.color_pallete[self.num_semantic_classes + 1] = np.array([128, 0, 128], dtype=np.uint8)
            pal = np.ascontiguousarray(self.color_pallete)
            _capi.check(self.lib.ssmap_set_palette(self._h, pal.ctypes.data, pal.shape[0]))
        self.maps: list[Optional[Dict[str, torch.Tensor]]] = [None] * self.n_envs
        self.map_bounds: list[Optional[np.ndarray]] = [None] * self.n_envs
        self._keep = None

    # ------------------------------------------------------------------ egocentric maps
    def set_egomap(self, egomap_small_side_m: float = 3.0, egomap_large_side_m: float = 10.0,
                   egomap_size=(128, 128), agent_draw_radius_m: float = 0.1, aux_num_bins: int = 12,
                   aux_radius_m: float = 0.4, aux_bin_m: float = 0.15, offset_for_cut: int = 150):
        """The ego-map constants the reference derives in its constructor (mapper.py:77-109)."""
        p2m = self.pixel2meter
        cfg = _capi.EgomapConfig()
        cfg.small_crop_px = int(egomap_small_side_m / p2m)
        cfg.large_crop_px = int(egomap_large_side_m / p2m)
        cfg.offset_for_cut = int(offset_for_cut)
        cfg.out_w, cfg.out_h = int(egomap_size[0]), int(egomap_size[1])
        cfg.agent_radius_px = int(agent_draw_radius_m / p2m)
        cfg.aux_bin_px = int(aux_bin_m / p2m)
        cfg.n_aux = int(aux_num_bins)
        radius_px = int(aux_radius_m / p2m)
        pts = [-np.array((radius_px * np.cos(np.deg2rad(d)), radius_px * np.sin(np.deg2rad(d))))
               for d in np.arange(0, 360, 360 / aux_num_bins)]
        pts = np.array(pts, dtype=np.int32).reshape(-1)            # truncation, like the reference's cast
        for i, vv in enumerate(pts):
            cfg.aux_points[i] = int(vv)
        n = self.num_semantic_classes
        for i in range(3):
            cfg.active_goal_color[i] = int(self.color_pallete[n][i])
            cfg.arrow_color[i] = int(self.color_pallete[n + 2][i])
        with torch.cuda.device(self.device):
            _capi.check(self.lib.ssmap_set_egomap(self._h, C.byref(cfg)))
        self._ego_shape = (cfg.out_h, cfg.out_w)
        self._ego_bins = cfg.n_aux

    def ego_maps(self, agent_pose, goal_sem_ids, aux_angle_onehot=None):
        """`SemanticMapper.ego_map` (mapper.py:876-998) for every env from the maps as they stand.
        Returns (ego_small, ego_large [E][h][w][3] u8, unc_small, unc_large [E][h][w] f32 or None)."""
        if getattr(self, "_ego_shape", None) is None:
            self.set_egomap()
        h, w = self._ego_shape
        ag = np.ascontiguousarray(agent_pose, dtype=np.float64).reshape(self.n_envs, 7)
        goals = np.ascontiguousarray(goal_sem_ids, dtype=np.int32).reshape(self.n_envs)
        aux = None
        if aux_angle_onehot is not None:
            onehot = np.asarray(aux_angle_onehot).reshape(self.n_envs, self._ego_bins)
            aux = np.ascontiguousarray(
                (np.array([255, 255, 255], dtype=np.uint8)[None, None, :] * onehot[:, :, None]).astype(np.uint8))
        small = torch.empty((self.n_envs, h, w, 3), dtype=torch.uint8, device=self.device)
        large = torch.empty_like(small)
        unc_s = unc_l = None
        if self.probablistic_mapping:
            unc_s = torch.empty((self.n_envs, h, w), dtype=torch.float32, device=self.device)
            unc_l = torch.empty_like(unc_s)
        with torch.cuda.device(self.device):
            _capi.check(self.lib.ssmap_ego_maps(
                self._h, ag.ctypes.data, goals.ctypes.data, aux.ctypes.data if aux is not None else None,
                small.data_ptr(), large.data_ptr(), unc_s.data_ptr() if unc_s is not None else None,
                unc_l.data_ptr() if unc_l is not None else None,
                C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return small, large, unc_s, unc_l

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.ssmap_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- schema (map_seg_to_schema, infer.py:133-138 / 336-341)
    def set_schema(self, lut: Sequence[int], pool_groups: Optional[Dict[int, Sequence[int]]] = None):
        """lut[c] = schema id of model class c.  pool_groups[g] (g in 3..8) lists the model
        classes pooled into semantic_max_prob for goal id g; default: every class mapped to g,
        ascending."""
        lut = np.ascontiguousarray(np.asarray(lut, dtype=np.int32))
        if pool_groups is None:
            pool_groups = {g: [int(k) for k in np.nonzero(lut == g)[0]] for g in range(3, 9)}
        ids = np.zeros((6, _capi.MAX_POOL), dtype=np.int32)
        cnt = np.zeros(6, dtype=np.int32)
        for g in range(3, 9):
            members = list(pool_groups.get(g, []))
            if len(members) > _capi.MAX_POOL:
                raise ValueError("too many classes pooled into one goal id")
            cnt[g - 3] = len(members)
            ids[g - 3, : len(members)] = members
        with torch.cuda.device(self.device):
            _capi.check(self.lib.ssmap_set_schema(self._h, lut.ctypes.data, lut.shape[0],
                                                  ids.ctypes.data, cnt.ctypes.data))
        self.schema_lut = lut

    # ---- path overlay (mapper.py:822-833)
    def path_append(self, e: int, cells):
        """Adds cells (flat map indices, host ints) to env e's trail; reset_env empties it."""
        cells = np.ascontiguousarray(np.asarray(cells, dtype=np.int32).reshape(-1))
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            _capi.check(self.lib.ssmap_path_append(self._h, int(e), cells.ctypes.data, int(cells.shape[0]),
                                                   C.c_void_p(stream)))

    def path_paint(self):
        """Stamps every env's trail into semantic_map / semantic_map_color (after `step`)."""
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            _capi.check(self.lib.ssmap_path_paint(self._h, C.c_void_p(stream)))

    # ---- naive-Bayes models of the stubborn rule (mapper.py:184-191, 502-515)
    def set_nb_model(self, goal_sem_id: int, model):
        """`model`: a fitted sklearn GaussianNB, or any object / dict with its `classes_`,
        `class_prior_`, `theta_` and `var_` (four features: total_view, cumulative_conf, max_conf,
        max_other_conf).  The constants are derived here exactly as sklearn's
        _joint_log_likelihood derives them."""
        get = (lambda k: model[k]) if isinstance(model, dict) else (lambda k: getattr(model, k))
        classes = np.ascontiguousarray(np.asarray(get("classes_")), dtype=np.int32)
        prior = np.asarray(get("class_prior_"), dtype=np.float64)
        theta = np.ascontiguousarray(np.asarray(get("theta_"), dtype=np.float64))
        var = np.ascontiguousarray(np.asarray(get("var_"), dtype=np.float64))
        n = classes.shape[0]
        if theta.shape != (n, 4) or var.shape != (n, 4) or prior.shape != (n,):
            raise ValueError("naive-Bayes model: theta_, var_ must be [n_classes, 4]")
        log_prior = np.ascontiguousarray([np.log(prior[i]) for i in range(n)], dtype=np.float64)
        konst = np.ascontiguousarray([-0.5 * np.sum(np.log(2.0 * np.pi * var[i, :])) for i in range(n)],
                                     dtype=np.float64)
        with torch.cuda.device(self.device):
            _capi.check(self.lib.ssmap_set_nb_model(self._h, int(goal_sem_id), int(n), classes.ctypes.data,
                                                    log_prior.ctypes.data, konst.ctypes.data,
                                                    theta.ctypes.data, var.ctypes.data))

    # ---- reset (mapper.py:193-269)
    def reset_env(self, e: int, map_bounds):
        b = np.asarray(map_bounds, dtype=np.float64)
        rows, cols = map_size(b, self.pixel2meter)
        dev, f32 = self.device, torch.float32
        m: Dict[str, torch.Tensor] = {
            "semantic_map": torch.zeros((rows, cols), dtype=torch.int32, device=dev),
            "semantic_map_color": torch.zeros((rows, cols, 3), dtype=torch.uint8, device=dev),
            "height_map": torch.full((rows, cols), -math.inf, dtype=f32, device=dev),
        }
        if self.probablistic_mapping:
            m["gt_map"] = torch.zeros((rows, cols), dtype=torch.int32, device=dev)
            m["unc_map"] = torch.full((rows, cols, 1), -0.5, dtype=f32, device=dev)
            if self.rule == "goal_decay":
                m["goal_map"] = torch.zeros((rows, cols), dtype=f32, device=dev)
            elif self.rule == "log_odds":
                m["log_odds_seg_map"] = torch.zeros((rows, cols, self.n_classes), dtype=f32, device=dev)
            elif self.rule in AVG_RULES:
                m["avg_prob_seg_map"] = torch.zeros((rows, cols, self.n_classes), dtype=f32, device=dev)
                m["avg_prob_weight_map"] = torch.zeros((rows, cols, 1), dtype=f32, device=dev)
            elif self.rule in ("hv", "prob_hv"):
                m["background_map"] = torch.zeros((rows, cols), dtype=torch.int32, device=dev)
                m["hits"] = torch.zeros((rows, cols), dtype=f32, device=dev)
                m["views"] = torch.zeros((rows, cols), dtype=f32, device=dev)
        else:
            m["gt_map"] = m["semantic_map"]
        if (self.probablistic_mapping and self.rule == "stubborn") or self.collect_nb_data:
            # mapper.py:244-250, 260-264
            m["total_view"] = torch.zeros((rows, cols), dtype=torch.int32, device=dev)
            m["cumulative_conf"] = torch.zeros((rows, cols), dtype=f32, device=dev)
            m["max_conf"] = torch.zeros((rows, cols), dtype=f32, device=dev)
            m["max_other_conf"] = torch.zeros((rows, cols), dtype=f32, device=dev)

        def ptr(name):
            t = m.get(name)
            return None if t is None else t.data_ptr()

        em = _capi.EnvMaps(map_h=rows, map_w=cols)
        em.lower[0], em.lower[1], em.lower[2] = float(b[0]), float(b[1]), float(b[2])
        em.semantic_map = ptr("semantic_map")
        em.semantic_map_color = ptr("semantic_map_color")
        em.height_map = ptr("height_map")
        em.gt_map = ptr("gt_map")
        em.unc_map = ptr("unc_map")
        em.prob_map = ptr("log_odds_seg_map") or ptr("avg_prob_seg_map")
        em.weight_map = ptr("avg_prob_weight_map")
        em.goal_map = ptr("goal_map")
        em.background_map = ptr("background_map")
        em.hits = ptr("hits")
        em.views = ptr("views")
        em.total_view = ptr("total_view")
        em.cumulative_conf = ptr("cumulative_conf")
        em.max_conf = ptr("max_conf")
        em.max_other_conf = ptr("max_other_conf")
        with torch.cuda.device(self.device):
            torch.cuda.current_stream().synchronize()   # fills above are stream ordered
            _capi.check(self.lib.ssmap_bind_env(self._h, int(e), C.byref(em)))
        self.maps[e] = m
        self.map_bounds[e] = b
        return m

    # ---- one batch of frames
    def step(self, depth: torch.Tensor, agent_pose, camera_pose, goal_sem_ids, *,
             logits: Optional[torch.Tensor] = None, probs: Optional[torch.Tensor] = None,
             temperature: float = 1.0, semantic: Optional[torch.Tensor] = None,
             semantic_unc: Optional[torch.Tensor] = None,
             semantic_max_prob: Optional[torch.Tensor] = None,
             semantic_gt: Optional[torch.Tensor] = None, static_scratch: bool = False, skip=None):
        """Enqueues stages 1-5 for all envs on the current stream.  `skip` (host bools [E]) names envs
        without a frame in this batch: their maps and state stay untouched, their results read
        found False, status 0, 0 updates.  With `static_scratch` the
        allocation-free entry point runs the step (per-step scratch in the library's file-scope
        device array; raises when this engine's scratch does not fit).  Tensors are [E, ...] on the
        engine's device, contiguous; `depth` is modified in place (sentinels zeroed).  Poses are
        host arrays [E,7] float64; goal_sem_ids host ints [E]."""
        E, H, W = self.n_envs, self.img_h, self.img_w

        def dev(t, dtype, shape, name):
            if t is None:
                return None
            if t.device != self.device or t.dtype != dtype or tuple(t.shape) != shape or not t.is_contiguous():
                raise ValueError(f"{name}: expected contiguous {dtype} {shape} on {self.device}, "
                                 f"got {t.dtype} {tuple(t.shape)} on {t.device}")
            return t.data_ptr()

        fr = _capi.Frames()
        fr.depth = dev(depth, torch.float32, (E, H, W), "depth")
        fr.logits = dev(logits, torch.float32, (E, self.n_classes, H, W), "logits")
        fr.probs = dev(probs, torch.float32, (E, self.n_classes, H, W), "probs")
        fr.temperature = float(temperature)
        fr.semantic = dev(semantic, torch.int32, (E, H, W), "semantic")
        fr.semantic_unc = dev(semantic_unc, torch.float32, (E, H, W), "semantic_unc")
        fr.semantic_max_prob = dev(semantic_max_prob, torch.float32, (E, H, W), "semantic_max_prob")
        fr.semantic_gt = dev(semantic_gt, torch.int32, (E, H, W), "semantic_gt")
        ap = np.ascontiguousarray(np.asarray(agent_pose, dtype=np.float64).reshape(E, 7))
        cp = np.ascontiguousarray(np.asarray(camera_pose, dtype=np.float64).reshape(E, 7))
        gi = np.ascontiguousarray(np.asarray(goal_sem_ids, dtype=np.int32).reshape(E))
        fr.agent_pose, fr.camera_pose, fr.goal_sem_id = ap.ctypes.data, cp.ctypes.data, gi.ctypes.data
        sk = None
        if skip is not None:
            sk = np.ascontiguousarray(np.asarray(skip).reshape(E).astype(np.uint8))
            fr.skip = sk.ctypes.data
        with torch.cuda.device(self.device):
            if self.params.dirty:
                pk = self.params.pack()
                _capi.check(self.lib.ssmap_set_params(self._h, C.byref(pk)))
                object.__setattr__(self.params, "dirty", False)
            stream = torch.cuda.current_stream().cuda_stream
            entry = self.lib.ssmap_step_noalloc if static_scratch else self.lib.ssmap_step
            _capi.check(entry(self._h, C.byref(fr), C.c_void_p(stream)))

    def stage1(self, logits: torch.Tensor, temperature: float, want_probs: bool = False):
        """The softmax / entropy / confidence tail of predict() for [E,C,H,W] logits."""
        E, H, W = self.n_envs, self.img_h, self.img_w
        assert logits.is_contiguous() and logits.dtype == torch.float32 and logits.device == self.device
        assert tuple(logits.shape) == (E, self.n_classes, H, W)
        sem = torch.empty((E, H, W), dtype=torch.int32, device=self.device)
        unc = torch.empty((E, H, W), dtype=torch.float32, device=self.device)
        maxp = torch.empty((E, H, W), dtype=torch.float32, device=self.device)
        probs = torch.empty_like(logits) if want_probs else None
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            _capi.check(self.lib.ssmap_stage1(self._h, logits.data_ptr(), float(temperature), sem.data_ptr(),
                                              unc.data_ptr(), maxp.data_ptr(),
                                              None if probs is None else probs.data_ptr(),
                                              C.c_void_p(stream)))
        out = {"semantic": sem, "semantic_unc": unc, "semantic_max_prob": maxp}
        if probs is not None:
            out["semantic_prob"] = probs
        return out

    def results(self):
        """Waits for the current stream; returns (found[E] bool, status[E] uint32, n_updates[E])."""
        E = self.n_envs
        found = np.zeros(E, dtype=np.int32)
        status = np.zeros(E, dtype=np.uint32)
        n_upd = np.zeros(E, dtype=np.int32)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            _capi.check(self.lib.ssmap_read_results(self._h, found.ctypes.data, status.ctypes.data,
                                                    n_upd.ctypes.data, C.c_void_p(stream)))
        return found.astype(bool), status, n_upd

    def gt_goal_query(self, in_disk: bool = True, anywhere: bool = True):
        """The two map questions of the shortest-path policy's found override (env.py:581-604), per
        env, after the last `step`: does gt_map hold the goal's id inside the found disk around the
        robot cell; does it hold it anywhere.  Returns (in_disk[E] or None, anywhere[E] or None) as
        host bool arrays; waits for the current stream."""
        E = self.n_envs
        a = np.zeros(E, dtype=np.int32) if in_disk else None
        b = np.zeros(E, dtype=np.int32) if anywhere else None
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            _capi.check(self.lib.ssmap_gt_goal_query(self._h, a.ctypes.data if in_disk else None,
                                                     b.ctypes.data if anywhere else None, C.c_void_p(stream)))
        return (a.astype(bool) if in_disk else None, b.astype(bool) if anywhere else None)

    def profile_begin(self, max_steps: int):
        _capi.check(self.lib.ssmap_profile_begin(self._h, int(max_steps)))

    def profile_read(self):
        """(ms per stage group summed over the recorded steps [5], number of steps)."""
        ms = np.zeros(5, dtype=np.float64)
        n = C.c_int32(0)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            _capi.check(self.lib.ssmap_profile_read(self._h, ms.ctypes.data, C.byref(n), C.c_void_p(stream)))
        return ms, int(n.value)

    def update_list(self, e: int, n_updates: int):
        """(pixels, cells) of env e's last step as host int arrays sorted by pixel index."""
        p = C.c_void_p()
        _capi.check(self.lib.ssmap_update_list(self._h, int(e), C.byref(p)))
        if n_updates == 0:
            return np.zeros(0, np.int64), np.zeros(0, np.int64)
        buf = torch.empty((n_updates, 2), dtype=torch.int32, device=self.device)
        torch.cuda.current_stream(self.device).synchronize()
        # device-to-device copy from the raw pointer through a uint8 view
        holder = _RawDeviceBuffer(p.value, n_updates * 8, self.device)
        buf.view(torch.uint8).reshape(-1).copy_(holder.tensor())
        host = buf.cpu().numpy().astype(np.int64)
        keep = (host[:, 0] & (1 << 29)) == 0     # dropped: another pixel owns the wrapped cell
        pix = host[:, 0] & ((1 << 29) - 1)
        order = np.argsort(pix, kind="stable")
        self.last_list_dropped = int((~keep).sum())
        return pix[order], host[order, 1]


class _RawDeviceBuffer:
    """Wraps a raw device pointer as a torch uint8 tensor via __cuda_array_interface__."""

    def __init__(self, ptr: int, nbytes: int, device):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False),
                                         "version": 2}
        self._device = device

    def tensor(self):
        return torch.as_tensor(self, device=self._device)
