"""Record golden vectors from the reference's own mapper.py (build container only).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

The reference file /root/reference/sem_objnav/obj_nav/mapper.py is imported by path, with
the four third-party modules this image lacks replaced by the restatements in
thirdparty_seams.py (parity at those seams is unpinned; see that file).  For every case
below a fresh reference `SemanticMapper` is driven through a short frame sequence and the
inputs, the per-frame update set (`valid_flat_idx`, captured by spying on
numpy.intersect1d) and the resulting map state are written to one .npz.  The reference
never travels: only these data files are committed.

Also checked here, once, at full frame size: the reference's rotate_vectors path
(numpy tensordot -> OpenBLAS) equals the oracle's fused k-ordered chain bit for bit.
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import thirdparty_seams  # noqa: E402
from tests import scenes  # noqa: E402

REF_MAPPER = "/root/reference/sem_objnav/obj_nav/mapper.py"


def load_reference():
    thirdparty_seams.install()
    spec = importlib.util.spec_from_file_location("ref_mapper", REF_MAPPER)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ----------------------------------------------------------------------------- cases
SCHEMA41_LUT = [0, 1, 2, 3, 11, 14, 18, 22, 10, 5, 30, 40]   # model class -> id of the 41-class schema
ROOM = dict(room=(0.0, 0.0, 4.5, 4.5), box=(2.6, 2.6, 3.3, 3.2, 0.7))
BOUNDS = [0.0, -0.2, 0.0, 4.5, 2.5, 4.5]
YAW_IN = np.pi + 0.75      # from (1,1) towards the box at ~(3,3)

CASES = [
    dict(name="direct_height_argmax", rule="height_map_argmax", prob=False, C=0, hw=(48, 64),
         frames=3, scene="room", start=(1.0, 1.0), step=0.4, yaw=YAW_IN),
    dict(name="avg_prob_c8_walk", rule="height_map_avg_prob", prob=True, C=8, hw=(60, 80),
         frames=5, scene="room", start=(1.0, 1.0), step=0.45, yaw=YAW_IN,
         attrs=dict(entropy_threshold=0.9)),
    dict(name="avg_prob_c40", rule="height_map_avg_prob", prob=True, C=40, hw=(30, 40),
         frames=3, scene="room", start=(1.2, 1.0), step=0.5, yaw=YAW_IN, T=3.2,
         attrs=dict(entropy_threshold=0.43759)),
    dict(name="avg_prob_plain_uniform", rule="avg_prob", prob=True, C=6, hw=(48, 64),
         frames=3, scene="uniform", start=(2.2, 2.2), step=0.0, yaw=0.3, turn=0.9),
    dict(name="w_avg_prob_c8", rule="height_map_w_avg_prob", prob=True, C=8, hw=(48, 64),
         frames=4, scene="room", start=(1.0, 1.0), step=0.45, yaw=YAW_IN,
         attrs=dict(entropy_threshold=0.8)),
    dict(name="w2_avg_prob_c12", rule="height_map_w2_avg_prob", prob=True, C=12, hw=(48, 64),
         frames=4, scene="room", start=(1.0, 1.0), step=0.45, yaw=YAW_IN, T=2.5,
         attrs=dict(entropy_threshold=0.482)),
    dict(name="log_odds_c8", rule="height_map_log_odds", prob=True, C=8, hw=(60, 80),
         frames=5, scene="room", start=(1.0, 1.0), step=0.45, yaw=YAW_IN,
         attrs=dict(entropy_threshold=0.3)),
    dict(name="log_odds_plain_uniform", rule="log_odds", prob=True, C=6, hw=(48, 64),
         frames=3, scene="uniform", start=(2.2, 2.2), step=0.0, yaw=1.1, turn=-0.7),
    dict(name="goal_decay_filter", rule="height_map_goal_decay", prob=True, C=8, hw=(120, 160),
         frames=6, scene="room", start=(1.0, 1.0), step=0.3, yaw=YAW_IN, filter_goal_prob=0.38158,
         speckle=0.005, sharp=12.0, box=(2.0, 2.0, 2.9, 2.8, 0.5),
         attrs=dict(goal_decay=0.8931, goal_mark=1.5408)),
    dict(name="goal_decay_nofilter", rule="height_map_goal_decay", prob=True, C=8, hw=(96, 128),
         frames=6, scene="room", start=(1.0, 1.0), step=0.3, yaw=YAW_IN,
         speckle=0.01, sharp=12.0, box=(2.0, 2.0, 2.9, 2.8, 0.5)),
    dict(name="argmax_filter", rule="height_map_argmax", prob=True, C=8, hw=(60, 80),
         frames=4, scene="room", start=(1.0, 1.0), step=0.5, yaw=YAW_IN, filter_goal_prob=0.53446),
    dict(name="argmax_plain", rule="argmax", prob=True, C=8, hw=(48, 64),
         frames=3, scene="room", start=(1.0, 1.0), step=0.5, yaw=YAW_IN),
    dict(name="hv_gate_filter", rule="height_map_hv", prob=True, C=8, hw=(96, 128),
         frames=6, scene="room", start=(1.0, 1.0), step=0.3, yaw=YAW_IN, filter_goal_prob=0.45,
         attrs=dict(hv_ratio=0.55, hv_min_views=1, hv_limit_obj_view_dist=3.2)),
    dict(name="hv_default", rule="height_map_hv", prob=True, C=8, hw=(60, 80),
         frames=4, scene="room", start=(1.0, 1.0), step=0.4, yaw=YAW_IN,
         attrs=dict(hv_min_views=2)),
    dict(name="prob_hv_gate", rule="height_map_prob_hv", prob=True, C=8, hw=(96, 128),
         frames=6, scene="room", start=(1.0, 1.0), step=0.3, yaw=YAW_IN,
         attrs=dict(hv_ratio=0.887, hv_min_views=1, hv_limit_obj_view_dist=4.13219)),
    dict(name="prob_hv_reject", rule="height_map_prob_hv", prob=True, C=8, hw=(96, 128),
         frames=6, scene="room", start=(1.0, 1.0), step=0.3, yaw=YAW_IN, sharp=1.5,
         attrs=dict(hv_ratio=0.95, hv_min_views=0.5, hv_limit_obj_view_dist=0.0)),
    # camera 0.45 m from the x=0 wall and looking at it: wall points land at col -4..-1
    dict(name="negative_wrap_avg", rule="height_map_avg_prob", prob=True, C=6, hw=(48, 64),
         frames=3, scene="room", start=(0.45, 2.0), step=0.0, yaw=np.pi / 2 + 0.05, turn=0.25),
    dict(name="negative_wrap_direct", rule="height_map_argmax", prob=False, C=0, hw=(48, 64),
         frames=2, scene="room", start=(2.0, 0.5), step=0.0, yaw=0.1, turn=-0.3),
    # yaw = 0 makes the roll entry of the rotation exactly zero; coarse depth quanta then
    # give exact height ties inside a cell (tie -> lowest pixel index)
    dict(name="height_ties", rule="height_map_argmax", prob=False, C=0, hw=(48, 64),
         frames=2, scene="room", start=(2.2, 3.6), step=0.3, yaw=0.0, quantum=0.25),
    # stubborn: argmax + goal statistics + the naive-Bayes veto; goal classes 3 and 9 pooled
    dict(name="stubborn_nb", rule="height_map_stubborn", prob=True, C=12, hw=(60, 80),
         frames=5, scene="room", start=(1.0, 1.0), step=0.45, yaw=YAW_IN, sharp=3.0,
         nb=True, goal_classes={3: [3, 9]}),
    dict(name="stubborn_nb_filter", rule="stubborn", prob=True, C=8, hw=(48, 64),
         frames=4, scene="room", start=(1.0, 1.0), step=0.5, yaw=YAW_IN, sharp=2.0,
         filter_goal_prob=0.3, nb=True, goal_classes={3: [3]}),
    # argmax collecting the training rows of those models
    dict(name="argmax_collect_nb", rule="height_map_argmax", prob=True, C=8, hw=(60, 80),
         frames=4, scene="room", start=(1.0, 1.0), step=0.5, yaw=YAW_IN,
         collect=True, goal_classes={3: [3]}),
    # stubborn while the rows are collected: statistics kept, the models not consulted (mapper.py:474-515)
    dict(name="stubborn_collect_nb", rule="height_map_stubborn", prob=True, C=8, hw=(48, 64),
         frames=4, scene="room", start=(1.0, 1.0), step=0.5, yaw=YAW_IN, sharp=2.0,
         nb=True, collect=True, goal_classes={3: [3]}),
    # the reference's 41- and 38-class schemas: goal ids 3, 11, 14, 18, 22, 10 (mapper.py:140-143),
    # no occupancy override, palette and path id sized by the schema
    dict(name="schema41_hv", rule="height_map_hv", prob=True, C=12, hw=(48, 64), n_schema=41, goal_index=2,
         lut=SCHEMA41_LUT, frames=4, scene="uniform", start=(2.2, 2.2), step=0.1, yaw=1.1, turn=-0.4,
         attrs=dict(hv_min_views=1, hv_ratio=0.3)),
    dict(name="schema41_avg_prob", rule="height_map_avg_prob", prob=True, C=12, hw=(48, 64), n_schema=41,
         goal_index=1, lut=SCHEMA41_LUT, frames=4, scene="uniform", start=(2.2, 2.2), step=0.1, yaw=0.3, turn=0.5,
         attrs=dict(entropy_threshold=0.9)),
    dict(name="schema38_argmax_filter", rule="height_map_argmax", prob=True, C=12, hw=(48, 64), n_schema=38,
         goal_index=5, lut=SCHEMA41_LUT[:11] + [37], frames=4, scene="uniform", start=(2.2, 2.2), step=0.1,
         yaw=2.0, turn=0.3, filter_goal_prob=0.3),
    # constructor parameters away from their defaults: cell size, height cut-off, found distance,
    # field of view, depth range (mapper.py:49-76)
    dict(name="ctor_variant_log_odds", rule="height_map_log_odds", prob=True, C=8, hw=(96, 128),
         frames=5, scene="room", start=(1.0, 1.0), step=0.3, yaw=YAW_IN, speckle=0.01, sharp=12.0,
         box=(2.0, 2.0, 2.9, 2.8, 0.5), hfov=79.0, depth_range=(0.3, 4.0),
         ctor=dict(pixel2meter=0.05, height_cutoff_m=0.3, success_dist_m=0.6), attrs=dict(entropy_threshold=0.4)),
    dict(name="ctor_variant_hv", rule="height_map_hv", prob=True, C=8, hw=(60, 80),
         frames=5, scene="room", start=(1.0, 1.0), step=0.3, yaw=YAW_IN, speckle=0.01, sharp=12.0,
         box=(2.0, 2.0, 2.9, 2.8, 0.5), hfov=45.0, depth_range=(0.4, 3.5),
         ctor=dict(pixel2meter=0.02, height_cutoff_m=-0.2, success_dist_m=1.5),
         attrs=dict(hv_min_views=1, hv_ratio=0.3)),
    dict(name="uniform_goal_decay", rule="height_map_goal_decay", prob=True, C=6, hw=(48, 64),
         frames=3, scene="uniform", start=(2.2, 2.2), step=0.0, yaw=2.0, turn=0.5,
         filter_goal_prob=0.3),
]


def build_frames(case):
    rng = np.random.default_rng(sum(ord(ch) * (i + 1) for i, ch in enumerate(case["name"])))
    h, w = case["hw"]
    c = case["C"]
    temp = case.get("T", 1.0)
    lut = None
    if c:
        lut = scenes.schema_lut(c) if c >= 37 else scenes.small_lut(c)
    if "lut" in case:
        lut = np.array(case["lut"], dtype=np.int64)
    poses = []
    yaw = case["yaw"]
    for i in range(case["frames"]):
        y = yaw + i * case.get("turn", 0.0)
        x = case["start"][0] - i * case["step"] * np.sin(y)
        z = case["start"][1] - i * case["step"] * np.cos(y)
        poses.append(scenes.make_pose(x, 0.0, z, y))
    frames = []
    for agent, cam in poses:
        if case["scene"] == "uniform":
            fr = scenes.uniform_frame(rng, h, w, c, temp, lut=lut)
        else:
            room = dict(ROOM)
            if "box" in case:
                room["box"] = case["box"]
            fr = scenes.room_frame(rng, h, w, c, temp, cam, lut=lut, sharp=case.get("sharp", 6.0),
                                   speckle=case.get("speckle", 0.03),
                                   quantum=case.get("quantum"), hfov_deg=case.get("hfov", 59.0),
                                   depth_range=tuple(case.get("depth_range", (0.5, 5.0))), **room)
        frames.append((fr, agent, cam))
    return frames, lut


def fit_nb_models(seed):
    """Fitted sklearn GaussianNB per goal id on synthetic feature rows {views, cumulative,
    max goal confidence, max other confidence}; label 1 = really the goal."""
    from sklearn.naive_bayes import GaussianNB
    rng = np.random.default_rng(seed)
    models = {}
    for goal in range(3, 9):
        n = 400
        views1 = rng.integers(1, 6, n)
        conf1 = np.clip(rng.normal(0.62, 0.12, n), 0.05, 1.0)
        views0 = rng.integers(1, 6, n)
        conf0 = np.clip(rng.normal(0.40, 0.10, n), 0.02, 1.0)
        x1 = np.stack([views1, views1 * conf1 * 0.9, conf1, np.clip(1 - conf1 - 0.1, 0.01, 1)], -1)
        x0 = np.stack([views0, views0 * conf0 * 0.8, conf0, np.clip(1 - conf0 - 0.2, 0.01, 1)], -1)
        models[goal] = GaussianNB().fit(np.concatenate([x0, x1]), np.concatenate([np.zeros(n), np.ones(n)]).astype(int))
    return models


def run_case(ref, case):
    frames, lut = build_frames(case)
    c = case["C"]
    nb_models = None
    kw = {}
    if case.get("nb"):
        # the reference unpickles nb_models/<model_name>.pkl in its constructor (mapper.py:184-191);
        # that file is opened but never read here: pickle.load is replaced for the call and hands
        # over models fitted by this script
        nb_models = fit_nb_models(len(case["name"]))
        kw["model_name"] = "emsanet"
        real_load = ref.pickle.load
        ref.pickle.load = lambda fh: nb_models
    if case.get("collect"):
        kw["collect_nb_data"] = True
    try:
        mapper = ref.SemanticMapper(
            num_semantic_classes=case.get("n_schema", 9), depth_hfov=case.get("hfov", 59.0),
            depth_range=tuple(case.get("depth_range", (0.5, 5.0))),
            probablistic_mapping=case["prob"], prob_aggregate=case["rule"],
            seg_model_n_classes=c or None, filter_goal_prob=case.get("filter_goal_prob"),
            **case.get("ctor", {}), **kw)
    finally:
        if case.get("nb"):
            ref.pickle.load = real_load
    if "goal_classes" in case:
        mapper.objnav_to_sem_pred = {int(k): list(v) for k, v in case["goal_classes"].items()}
    for k, v in case.get("attrs", {}).items():
        setattr(mapper, k, v)
    if lut is not None:
        mapper.map_seg_to_schema = lambda ids: lut[ids]
    mapper.reset(np.array(BOUNDS, dtype=np.float64))

    out = {}
    meta = dict(name=case["name"], rule=case["rule"], prob=case["prob"], C=c,
                hw=list(case["hw"]), n_frames=len(frames), bounds=BOUNDS,
                filter_goal_prob=case.get("filter_goal_prob"), attrs=case.get("attrs", {}),
                collect_nb_data=bool(case.get("collect")),
                n_schema=case.get("n_schema", 9), goal_index=case.get("goal_index", 0),
                hfov=case.get("hfov", 59.0), depth_range=list(case.get("depth_range", (0.5, 5.0))),
                ctor=case.get("ctor", {}),
                goal_classes={str(k): list(v) for k, v in case.get("goal_classes", {}).items()},
                nb_models=None if nb_models is None else {
                    str(g): dict(classes_=[int(v) for v in m.classes_],
                                 class_prior_=[float(v) for v in m.class_prior_],
                                 theta_=[[float(v) for v in r] for r in m.theta_],
                                 var_=[[float(v) for v in r] for r in m.var_])
                    for g, m in nb_models.items()},
                lut=None if lut is None else [int(v) for v in lut],
                numpy=np.__version__)
    for i, (fr, agent, cam) in enumerate(frames):
        obs = scenes.make_obs(fr, agent, cam, goal_index=case.get("goal_index", 0))
        # keep only the observation keys this rule reads (mapper.py:361-727)
        uses_probs = any(t in case["rule"] for t in ("avg_prob", "log_odds"))
        keeps_probs = bool(case.get("nb") or case.get("collect"))   # goal statistics read semantic_prob
        if case["prob"] and not uses_probs and not keeps_probs:
            obs.pop("semantic_prob", None)
            if "prob_hv" in case["rule"]:   # mapper.py:668 only transposes it
                obs["semantic_prob"] = np.zeros((1,) + obs["depth"].shape, dtype=np.float32)
        if uses_probs:
            for key in ("semantic", "semantic_unc", "semantic_max_prob"):
                obs.pop(key, None)
        for key in ("depth", "semantic", "semantic_gt", "semantic_prob", "semantic_unc",
                    "semantic_max_prob"):
            if key in obs and not (key == "semantic_prob" and not (uses_probs or keeps_probs)):
                out[f"f{i}_in_{key}"] = obs[key].copy()
        out[f"f{i}_in_agent"] = agent
        out[f"f{i}_in_camera"] = cam
        spied = []
        real = np.intersect1d

        def spy(a, b, *args, **kw):
            res = real(a, b, *args, **kw)
            spied.append(res)
            return res

        np.intersect1d = spy
        try:
            _, found = mapper.update_map(obs, None)
        finally:
            np.intersect1d = real
        out[f"f{i}_update_pixels"] = spied[1].astype(np.int64)
        if i == 0:
            out[f"f{i}_depth_after"] = obs["depth"]
        out[f"f{i}_found"] = np.array(1 if found else 0, dtype=np.int8)
        out[f"f{i}_semantic_map"] = mapper.semantic_map.astype(np.uint8)
        # float state is recorded per frame only as a float64 sum (cheap drift check);
        # the full arrays are recorded after the last frame
        for name in ("height_map", "unc_map", "goal_map", "hits", "views"):
            val = getattr(mapper, name, None)
            if val is not None:
                finite = np.isfinite(val)
                out[f"f{i}_sum_{name}"] = np.array(
                    [val[finite].astype(np.float64).sum(), float(finite.sum())])
    for name in ("semantic_map", "semantic_map_color", "height_map", "gt_map", "unc_map",
                 "goal_map", "log_odds_seg_map", "avg_prob_seg_map", "avg_prob_weight_map",
                 "background_map", "hits", "views", "total_view", "cumulative_conf", "max_conf",
                 "max_other_conf"):
        val = getattr(mapper, name, None)
        if val is not None:
            out[f"final_{name}"] = val.copy()
    if case.get("collect"):
        for key in ("goal", "non_goal"):
            rows = mapper.eps_nb_data[key]
            out[f"final_nb_{key}"] = np.concatenate(rows) if rows else np.zeros((0, 4))
            out[f"final_nb_{key}_chunks"] = np.array([len(r) for r in rows], dtype=np.int64)
    out["meta"] = np.array(json.dumps(meta))
    return out


def check_rotation_seam(ref):
    """rotate_vectors (tensordot/OpenBLAS) == fused chain of oracle/geometry.c, 640x480."""
    from oracle import mapper_oracle as mo
    rng = np.random.default_rng(7)
    depth = rng.uniform(0.6, 4.9, (480, 640)).astype(np.float32)
    agent, cam = scenes.make_pose(7.2, 0.0, 7.2, 1.234)
    rows, cols, heights, valid = mo.project_frame(depth.copy(), 59.0, (0.5, 5.0), cam,
                                                  np.zeros(3), 0.03)
    pc, _ = ref.pointcloud(depth, 59.0)
    q = thirdparty_seams.from_float_array(cam[3:]) * thirdparty_seams.from_rotation_vector(
        np.array([np.pi, 0, 0]))
    world = thirdparty_seams.rotate_vectors(q, pc.reshape(-1, 3)) + cam[:3]
    assert np.array_equal(world[:, 1], heights), "height mismatch vs tensordot path"
    cells = np.floor(np.stack([world[:, 2], world[:, 0]], -1) / 0.03).astype(np.int64)
    assert np.array_equal(cells[:, 0], rows) and np.array_equal(cells[:, 1], cols)
    print("rotation seam: tensordot path == fused chain on 307200 points")


def main():
    ref = load_reference()
    check_rotation_seam(ref)
    total = 0
    only = set(sys.argv[1:])          # case names; none = all
    for case in CASES:
        if only and case["name"] not in only:
            continue
        out = run_case(ref, case)
        path = os.path.join(HERE, case["name"] + ".npz")
        np.savez_compressed(path, **out)
        size = os.path.getsize(path)
        total += size
        n_found = sum(int(out[f"f{i}_found"]) for i in range(case["frames"]))
        k = [len(out[f"f{i}_update_pixels"]) for i in range(case["frames"])]
        print(f"{case['name']:28s} {size/1024:8.1f} KiB  K={k} found={n_found}")
    print(f"total {total/1024:.1f} KiB")


if __name__ == "__main__":
    main()
