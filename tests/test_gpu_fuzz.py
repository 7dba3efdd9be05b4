"""Seeded random scenarios, HIP engine (through the C ABI) against the oracle: image sizes, map
bounds, poses near and beyond the map edges (negative index wrap, cells above the upper bounds),
pitch, depth sentinels, depths beyond depth_max (frames that leave the predicted cell box), flat
depth images (thousands of pixels in one cell), structured and uncorrelated depth.

Bar: update set, height map, ground-truth map and labels bit-exact; fused probabilities within
rtol 1e-5 / atol 1e-6.  numpy's IndexError (a selected cell index below -dim) is reported by the
engine on exactly the frames on which the oracle raises; such a scenario ends there."""
import os

import numpy as np
import pytest
import torch

from oracle import mapper_oracle as mo
from tests import parity, scenes

pytestmark = pytest.mark.gpu

# SSMAP_FUZZ_SCALE=8 runs eight times as many seeds of every generator (a wider one-off sweep)
_SCALE = max(1, int(os.environ.get("SSMAP_FUZZ_SCALE", "1")))


def _dev(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


def _scenario(seed):
    rng = np.random.default_rng(1000 + seed)
    # widths that are no multiple of four take the engine's one-pixel-at-a-time geometry variant
    h, w = [(24, 32), (60, 80), (96, 128), (120, 160), (50, 68), (240, 320), (45, 67), (33, 50)][seed % 8]
    c = int(rng.integers(3, 10))
    span_x, span_z = rng.uniform(3.0, 9.0, 2)
    lower = rng.uniform(-3.0, 3.0, 3)
    bounds = np.array([lower[0], lower[1], lower[2], lower[0] + span_x, lower[1] + 2.5, lower[2] + span_z])
    # agent anywhere from one metre outside the map to its middle
    ax = rng.uniform(lower[0] - 1.0, lower[0] + span_x + 1.0)
    az = rng.uniform(lower[2] - 1.0, lower[2] + span_z + 1.0)
    pitch = rng.choice([scenes.CAM_PITCH, 0.0, -0.8, 0.3])
    kind = ["uniform", "flat", "beyond", "near", "broken", "infinite"][int(rng.integers(0, 6))]
    rule = ["height_map_avg_prob", "height_map_log_odds", "avg_prob"][int(rng.integers(0, 3))]
    return dict(rng=rng, hw=(h, w), c=c, bounds=bounds, agent=(ax, az), pitch=float(pitch), kind=kind,
                rule=rule, floor=float(lower[1]))


def _depth(sc, rng):
    h, w = sc["hw"]
    d = rng.uniform(0.6, 4.9, (h, w)).astype(np.float32)
    if sc["kind"] == "flat":          # one depth for the whole image: long runs of pixels per cell
        d[:] = np.float32(rng.uniform(0.7, 4.5))
        d[rng.random((h, w)) < 0.2] = np.float32(rng.uniform(0.7, 4.5))
    elif sc["kind"] == "beyond":      # some pixels past depth_max, outside the predicted cell box
        far = rng.random((h, w)) < 0.03
        d[far] = rng.uniform(5.5, 12.0, int(far.sum())).astype(np.float32)
    elif sc["kind"] == "near":        # everything within a metre: a handful of cells
        d = rng.uniform(0.5, 1.0, (h, w)).astype(np.float32)
    elif sc["kind"] in ("broken", "infinite"):   # a sensor that reports NaN, zeros, negative depths ...
        bad = rng.random((h, w))
        d[bad < 0.02] = np.float32(np.nan)
        d[(bad >= 0.04) & (bad < 0.06)] = np.float32(-np.inf)
        d[(bad >= 0.06) & (bad < 0.08)] = np.float32(-1.5)
        d[(bad >= 0.08) & (bad < 0.10)] = np.float32(0.0)
        if sc["kind"] == "infinite":             # ... and +inf: cell indices that numpy casts to INT64_MIN
            d[(bad >= 0.02) & (bad < 0.04)] = np.float32(np.inf)
            d[(bad >= 0.10) & (bad < 0.11)] = np.float32(3.0e38)
    sentinel = rng.random((h, w))
    d[sentinel < 0.01] = np.float32(0.5)
    d[sentinel > 0.99] = np.float32(5.0)
    return d


def _degenerate(probs, rng):
    """One-hot rows (exact zeros and ones), all-zero rows and denormal-small rows among the
    probability vectors: the log-odds of these are infinities, the averages divide by zero."""
    c, h, w = probs.shape
    kind = rng.random((h, w))
    hot = rng.integers(0, c, (h, w))
    onehot = (np.arange(c)[:, None, None] == hot[None]).astype(np.float32)
    probs = np.where(kind[None] < 0.2, onehot, probs)
    probs = np.where((kind[None] >= 0.2) & (kind[None] < 0.25), np.float32(0.0), probs)
    probs = np.where((kind[None] >= 0.25) & (kind[None] < 0.3), np.float32(1e-40) * onehot, probs)
    return np.ascontiguousarray(probs, dtype=np.float32)


@pytest.mark.parametrize("degenerate", [False, True])
@pytest.mark.parametrize("seed", range(36 * _SCALE))
def test_random_scenario_matches_oracle(seed, degenerate):
    import semantic_search_amd as ssa
    sc = _scenario(seed)
    rng, (h, w), c = sc["rng"], sc["hw"], sc["c"]
    lut = scenes.small_lut(c)
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate=sc["rule"], seg_model_n_classes=c)
    ref = mo.OracleMapper(schema_lut=lut, path_overlay=False, **kw)
    eng = ssa.MapEngine(1, (h, w), max_map_hw=ssa.map_size(sc["bounds"], 0.03), **kw)
    eng.set_schema(lut)
    ref.reset(sc["bounds"])
    maps = eng.reset_env(0, sc["bounds"])
    name = "log_odds_seg_map" if "log_odds" in sc["rule"] else "avg_prob_seg_map"
    for i in range(3):
        frame = scenes.uniform_frame(rng, h, w, c, 1.0, lut=lut)
        frame["depth"] = _depth(sc, rng)
        if degenerate:
            frame["semantic_prob"] = _degenerate(frame["semantic_prob"], rng)
        agent, cam = scenes.make_pose(sc["agent"][0] + 0.2 * i, sc["floor"], sc["agent"][1], rng.uniform(0, 2 * np.pi),
                                      pitch=sc["pitch"])
        obs = scenes.make_obs(frame, agent, cam, goal_index=0)
        obs["depth"] = frame["depth"].copy()
        raised = False
        try:
            with np.errstate(all="ignore"):
                ref.update(obs)
        except IndexError:
            raised = True
        before = {k: t.clone() for k, t in maps.items()}
        eng.step(_dev(frame["depth"][None], np.float32), agent[None], cam[None], [3],
                 probs=_dev(frame["semantic_prob"][None], np.float32),
                 semantic_gt=_dev(frame["semantic_gt"][None], np.int32))
        found, status, n_upd = eng.results()
        # numpy raises only when the offending pixel is the tallest of its cell: so does the engine
        assert bool(status[0] & 1) == raised, f"frame {i}: numpy raised IndexError: {raised}, the engine reported: {bool(status[0] & 1)}"
        if raised:
            # ... before its first map write (mapper.py:338): the frame leaves every map as it was
            assert n_upd[0] == 0
            for k, t in maps.items():
                assert torch.equal(t.view(torch.uint8), before[k].view(torch.uint8)), f"frame {i}: {k} changed by a frame that raises"
            eng.close()
            return
        pix, _ = eng.update_list(0, int(n_upd[0]))
        assert np.array_equal(np.unique(pix), np.sort(ref.last_update_pixels)), f"frame {i}: update set ({sc['kind']})"
        assert np.array_equal(maps["height_map"].cpu().numpy(), ref.height_map), f"frame {i}: height_map"
    assert np.array_equal(maps["gt_map"].cpu().numpy().view(np.uint32), ref.gt_map)
    got, want = maps[name].cpu().numpy(), getattr(ref, name)
    if "log_odds" in sc["rule"]:
        parity.close_rows(got, want, name)          # relative to the cell's largest finite |log-odds|
    else:
        parity.close(got, want, name)
    sem = maps["semantic_map"].cpu().numpy().view(np.uint32)
    parity.labels_agree(sem, ref.semantic_map, want)
    eng.close()


def _degenerate_planes(frame, rng):
    """NaN, infinite, negative and above-one values in the uncertainty and confidence planes
    (a segmentation wrapper that misbehaves): every comparison and sum must go numpy's way."""
    for key in ("semantic_unc", "semantic_max_prob"):
        plane = frame[key].copy()
        bad = rng.random(plane.shape)
        plane[bad < 0.01] = np.float32(np.nan)
        plane[(bad >= 0.01) & (bad < 0.02)] = np.float32(np.inf)
        plane[(bad >= 0.02) & (bad < 0.03)] = np.float32(-0.75)
        plane[(bad >= 0.03) & (bad < 0.04)] = np.float32(1.5)
        frame[key] = plane


@pytest.mark.parametrize("seed", range(16 * _SCALE))
def test_random_scenario_label_rules(seed):
    """The same scenario generator for the label rules that have exact wrap semantics (argmax with
    and without the goal filter, goal_decay, direct mapping): every integer map bit-exact."""
    import semantic_search_amd as ssa
    sc = _scenario(100 + seed)
    rng, (h, w), c = sc["rng"], sc["hw"], sc["c"]
    lut = scenes.small_lut(c)
    rule, prob, filt = [("height_map_argmax", True, None), ("height_map_argmax", True, 0.4),
                        ("height_map_goal_decay", True, 0.35), ("height_map_argmax", False, None)][seed % 4]
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=prob,
              prob_aggregate=rule, seg_model_n_classes=c, filter_goal_prob=filt)
    ref = mo.OracleMapper(schema_lut=lut, path_overlay=False, **kw)
    eng = ssa.MapEngine(1, (h, w), max_map_hw=ssa.map_size(sc["bounds"], 0.03), **kw)
    eng.set_schema(lut)
    ref.reset(sc["bounds"])
    maps = eng.reset_env(0, sc["bounds"])
    for i in range(3):
        frame = scenes.uniform_frame(rng, h, w, c, 1.0, lut=lut)
        frame["depth"] = _depth(sc, rng)
        if seed % 3 == 2:
            _degenerate_planes(frame, rng)
        agent, cam = scenes.make_pose(sc["agent"][0] + 0.2 * i, sc["floor"], sc["agent"][1], rng.uniform(0, 2 * np.pi),
                                      pitch=sc["pitch"])
        obs = scenes.make_obs(frame, agent, cam, goal_index=0)
        obs["depth"] = frame["depth"].copy()
        raised = False
        try:
            with np.errstate(all="ignore"):
                found_ref = ref.update(obs)
        except IndexError:
            raised = True
        extra = {}
        if prob:
            extra = dict(semantic_unc=_dev(frame["semantic_unc"][None], np.float32),
                         semantic_max_prob=_dev(frame["semantic_max_prob"][None], np.float32),
                         semantic_gt=_dev(frame["semantic_gt"][None], np.int32))
        eng.step(_dev(frame["depth"][None], np.float32), agent[None], cam[None], [3],
                 semantic=_dev(frame["semantic"][None], np.int32), **extra)
        found, status, n_upd = eng.results()
        assert bool(status[0] & 1) == raised, f"frame {i}: numpy raised IndexError: {raised}, the engine reported: {bool(status[0] & 1)}"
        if raised:
            eng.close()
            return
        assert bool(found[0]) == bool(found_ref), f"frame {i}: found"
        assert np.array_equal(maps["semantic_map"].cpu().numpy().view(np.uint32), ref.semantic_map), f"frame {i} ({rule})"
    assert np.array_equal(maps["height_map"].cpu().numpy(), ref.height_map)
    assert np.array_equal(maps["semantic_map_color"].cpu().numpy(), ref.semantic_map_color)
    if prob:
        assert np.array_equal(maps["gt_map"].cpu().numpy().view(np.uint32), ref.gt_map)
        assert np.array_equal(maps["unc_map"].cpu().numpy(), ref.unc_map, equal_nan=True)
    if "goal_decay" in rule:
        assert np.array_equal(maps["goal_map"].cpu().numpy(), ref.goal_map)
    eng.close()


@pytest.mark.parametrize("seed", range(40 * _SCALE))
def test_random_scenario_hit_view_rules(seed):
    """height_map_hv / height_map_prob_hv: background, hits, views and labels bit-exact, including
    update pixels that share a cell through the negative-index wrap (numpy applies every statement
    of the rule with its own 'last write wins')."""
    import semantic_search_amd as ssa
    sc = _scenario(200 + seed)
    rng, (h, w), c = sc["rng"], sc["hw"], sc["c"]
    lut = scenes.small_lut(c)
    rule = ["height_map_hv", "height_map_prob_hv"][seed % 2]
    filt = 0.4 if (rule == "height_map_hv" and seed % 3 == 0) else None
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate=rule, seg_model_n_classes=c, filter_goal_prob=filt)
    ref = mo.OracleMapper(schema_lut=lut, path_overlay=False, **kw)
    eng = ssa.MapEngine(1, (h, w), max_map_hw=ssa.map_size(sc["bounds"], 0.03), **kw)
    for o in (ref, eng.params):
        o.hv_min_views = 1
        o.hv_ratio = 0.3
        o.hv_limit_obj_view_dist = 3.0 if seed % 4 < 2 else 0.0
    eng.set_schema(lut)
    ref.reset(sc["bounds"])
    maps = eng.reset_env(0, sc["bounds"])
    for i in range(3):
        frame = scenes.uniform_frame(rng, h, w, c, 1.0, lut=lut)
        frame["depth"] = _depth(sc, rng)
        if seed % 3 == 2:
            _degenerate_planes(frame, rng)
        agent, cam = scenes.make_pose(sc["agent"][0] + 0.2 * i, sc["floor"], sc["agent"][1], rng.uniform(0, 2 * np.pi),
                                      pitch=sc["pitch"])
        obs = scenes.make_obs(frame, agent, cam, goal_index=0)
        obs["depth"] = frame["depth"].copy()
        raised = False
        try:
            with np.errstate(all="ignore"):
                found_ref = ref.update(obs)
        except IndexError:
            raised = True
        eng.step(_dev(frame["depth"][None], np.float32), agent[None], cam[None], [3],
                 semantic=_dev(frame["semantic"][None], np.int32),
                 semantic_unc=_dev(frame["semantic_unc"][None], np.float32),
                 semantic_max_prob=_dev(frame["semantic_max_prob"][None], np.float32),
                 semantic_gt=_dev(frame["semantic_gt"][None], np.int32))
        found, status, n_upd = eng.results()
        assert bool(status[0] & 1) == raised, f"frame {i}: numpy raised IndexError: {raised}, the engine reported: {bool(status[0] & 1)}"
        if raised:
            eng.close()
            return
        assert bool(found[0]) == bool(found_ref), f"frame {i}: found"
        for nm in ("semantic_map", "background_map"):
            assert np.array_equal(maps[nm].cpu().numpy().view(np.uint32), getattr(ref, nm)), f"frame {i}: {nm}"
        for nm in ("hits", "views", "height_map"):
            assert np.array_equal(maps[nm].cpu().numpy(), getattr(ref, nm), equal_nan=True), f"frame {i}: {nm}"
    assert np.array_equal(maps["unc_map"].cpu().numpy(), ref.unc_map, equal_nan=True)
    eng.close()


@pytest.mark.parametrize("seed", range(12 * _SCALE))
def test_random_scenario_stubborn_rule(seed):
    """height_map_stubborn under the random scenarios: labels and view counts bit-exact, the
    float statistics within 1e-6 (sums of a few float32 probabilities in the same order)."""
    import semantic_search_amd as ssa
    from tests import golden_util as gu
    sc = _scenario(300 + seed)
    rng, (h, w), c = sc["rng"], sc["hw"], max(sc["c"], 5)
    lut = scenes.small_lut(c)
    model = gu.GoldenCase("stubborn_nb").mapper_kwargs()["nb_models"][3]
    goal_classes = {3: [3, 4] if seed % 2 else [3]}
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate="height_map_stubborn", seg_model_n_classes=c)
    ref = mo.OracleMapper(schema_lut=lut, path_overlay=False, nb_models={3: model}, objnav_to_sem_pred=goal_classes, **kw)
    eng = ssa.MapEngine(1, (h, w), max_map_hw=ssa.map_size(sc["bounds"], 0.03), **kw)
    eng.set_schema(lut, {g: goal_classes.get(g, []) for g in range(3, 9)})
    eng.set_nb_model(3, model)
    ref.reset(sc["bounds"])
    maps = eng.reset_env(0, sc["bounds"])
    for i in range(3):
        frame = scenes.uniform_frame(rng, h, w, c, 1.0, lut=lut)
        frame["depth"] = _depth(sc, rng)
        if seed % 3 == 2:      # NaN and infinite confidences: the running maxima keep numpy's NaN rule
            frame["semantic_prob"] = _degenerate(frame["semantic_prob"], rng)
            nan_px = rng.random((h, w)) < 0.02
            frame["semantic_prob"][:, nan_px] = np.float32(np.nan)
            _degenerate_planes(frame, rng)
        agent, cam = scenes.make_pose(sc["agent"][0] + 0.2 * i, sc["floor"], sc["agent"][1], rng.uniform(0, 2 * np.pi),
                                      pitch=sc["pitch"])
        obs = scenes.make_obs(frame, agent, cam, goal_index=0)
        obs["depth"] = frame["depth"].copy()
        raised = False
        try:
            with np.errstate(all="ignore"):
                ref.update(obs)
        except IndexError:
            raised = True
        eng.step(_dev(frame["depth"][None], np.float32), agent[None], cam[None], [3],
                 probs=_dev(frame["semantic_prob"][None], np.float32),
                 semantic=_dev(frame["semantic"][None], np.int32),
                 semantic_unc=_dev(frame["semantic_unc"][None], np.float32),
                 semantic_max_prob=_dev(frame["semantic_max_prob"][None], np.float32),
                 semantic_gt=_dev(frame["semantic_gt"][None], np.int32))
        found, status, n_upd = eng.results()
        assert bool(status[0] & 1) == raised, f"frame {i}: numpy raised IndexError: {raised}, the engine reported: {bool(status[0] & 1)}"
        if raised:
            eng.close()
            return
        assert np.array_equal(maps["semantic_map"].cpu().numpy().view(np.uint32), ref.semantic_map), f"frame {i}"
        assert np.array_equal(maps["total_view"].cpu().numpy().view(np.uint32), ref.total_view), f"frame {i}"
    for nm in ("cumulative_conf", "max_conf", "max_other_conf"):
        assert np.allclose(maps[nm].cpu().numpy(), getattr(ref, nm), rtol=1e-6, atol=1e-7, equal_nan=True), nm
    eng.close()


@pytest.mark.parametrize("seed", range(24 * _SCALE))
def test_random_rooms_goal_decay(seed):
    """goal_decay on ray-cast rooms with a goal-class box: the box top is a blob of goal cells that
    survives the erosion, so the dilated goal mask, the decay of the other cells and (with map
    bounds that start inside the room) the wrapped twins are all in play.  Bit-exact.  (On the 96x128
    and 120x160 frames goal cells survive the erosion in nearly every seed; the smaller frames are too
    sparse for that and exercise the empty-mask shortcut instead.)"""
    import semantic_search_amd as ssa
    rng = np.random.default_rng(5000 + seed)
    h, w = [(96, 128), (120, 160), (60, 80), (45, 67)][seed % 4]
    c = 8
    lut = scenes.small_lut(c)
    filt = [None, 0.35][seed % 2]
    # lower bound inside the room on some seeds: part of the room lies at negative cell indices
    lower = np.array([rng.uniform(-0.5, 1.5), -0.2, rng.uniform(-0.5, 1.5)]) if seed % 3 else np.array([0.0, -0.2, 0.0])
    bounds = np.concatenate([lower, lower + np.array([rng.uniform(3.5, 6.0), 2.7, rng.uniform(3.5, 6.0)])])
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate="height_map_goal_decay", seg_model_n_classes=c, filter_goal_prob=filt)
    ref = mo.OracleMapper(schema_lut=lut, path_overlay=False, **kw)
    eng = ssa.MapEngine(1, (h, w), max_map_hw=ssa.map_size(bounds, 0.03), **kw)
    for o in (ref, eng.params):
        o.goal_decay = 0.8931
        o.goal_mark = 1.5408
    eng.set_schema(lut)
    ref.reset(bounds)
    maps = eng.reset_env(0, bounds)
    bx, bz = rng.uniform(1.9, 2.3, 2)
    box = (bx, bz, bx + rng.uniform(0.5, 0.9), bz + rng.uniform(0.5, 0.9), rng.uniform(0.3, 0.6))
    # the agent looks along (-sin yaw, -cos yaw): from the room's low corner towards the box
    ax, az, yaw = rng.uniform(0.8, 1.2), rng.uniform(0.8, 1.2), np.pi + rng.uniform(0.6, 0.95)
    for i in range(4):
        agent, cam = scenes.make_pose(ax + 0.1 * i, 0.0, az + 0.1 * i, yaw + 0.03 * i)
        frame = scenes.room_frame(rng, h, w, c, 1.0, cam, lut=lut, box=box, speckle=0.01, sharp=12.0)
        obs = scenes.make_obs(frame, agent, cam, goal_index=0)
        raised = False
        try:
            found_ref = ref.update(obs)
        except IndexError:
            raised = True
        eng.step(_dev(frame["depth"][None], np.float32), agent[None], cam[None], [3],
                 semantic=_dev(frame["semantic"][None], np.int32),
                 semantic_unc=_dev(frame["semantic_unc"][None], np.float32),
                 semantic_max_prob=_dev(frame["semantic_max_prob"][None], np.float32),
                 semantic_gt=_dev(frame["semantic_gt"][None], np.int32))
        found, status, n_upd = eng.results()
        assert bool(status[0] & 1) == raised, f"frame {i}: numpy raised IndexError: {raised}, the engine reported: {bool(status[0] & 1)}"
        if raised:
            eng.close()
            return
        assert bool(found[0]) == bool(found_ref), f"frame {i}: found"
        assert np.array_equal(maps["goal_map"].cpu().numpy(), ref.goal_map), f"frame {i}: goal_map"
        assert np.array_equal(maps["semantic_map"].cpu().numpy().view(np.uint32), ref.semantic_map), f"frame {i}"
    assert np.array_equal(maps["height_map"].cpu().numpy(), ref.height_map)
    assert np.array_equal(maps["semantic_map_color"].cpu().numpy(), ref.semantic_map_color)
    eng.close()


@pytest.mark.parametrize("seed", range(30 * _SCALE))
def test_random_rooms_found_decision(seed):
    """Ray-cast rooms with a goal-class box and an agent that walks up to it: the found decision of
    every frame (disk around the robot, entropy threshold or hit/view ratio) equals the oracle's,
    labels bit-exact (probability rules: except where the two leading classes tie)."""
    import semantic_search_amd as ssa
    rng = np.random.default_rng(7000 + seed)
    rule = ["height_map_log_odds", "height_map_avg_prob", "height_map_hv", "height_map_prob_hv",
            "height_map_argmax", "height_map_w2_avg_prob"][seed % 6]
    h, w = [(96, 128), (120, 160)][seed % 2]
    c = 8
    lut = scenes.small_lut(c)
    lower = np.array([rng.uniform(-0.5, 0.8), -0.2, rng.uniform(-0.5, 0.8)]) if seed % 3 else np.array([0.0, -0.2, 0.0])
    bounds = np.concatenate([lower, lower + np.array([rng.uniform(4.0, 6.0), 2.7, rng.uniform(4.0, 6.0)])])
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate=rule, seg_model_n_classes=c)
    ref = mo.OracleMapper(schema_lut=lut, path_overlay=False, **kw)
    eng = ssa.MapEngine(1, (h, w), max_map_hw=ssa.map_size(bounds, 0.03), **kw)
    threshold = float(rng.uniform(0.2, 0.6))
    for o in (ref, eng.params):
        if "hv" in rule:
            o.hv_min_views = 1
            o.hv_ratio = 0.4
        elif "argmax" not in rule:
            o.entropy_threshold = threshold
    eng.set_schema(lut)
    ref.reset(bounds)
    maps = eng.reset_env(0, bounds)
    bx, bz = rng.uniform(1.9, 2.3, 2)
    box = (bx, bz, bx + rng.uniform(0.5, 0.9), bz + rng.uniform(0.5, 0.9), rng.uniform(0.3, 0.6))
    ax, az, yaw = rng.uniform(0.8, 1.1), rng.uniform(0.8, 1.1), np.pi + rng.uniform(0.65, 0.9)
    prob_rule = "avg_prob" in rule or "log_odds" in rule
    name = "log_odds_seg_map" if "log_odds" in rule else "avg_prob_seg_map"
    seen = []
    for i in range(5):
        agent, cam = scenes.make_pose(ax + 0.15 * i, 0.0, az + 0.15 * i, yaw + 0.02 * i)
        frame = scenes.room_frame(rng, h, w, c, 1.0, cam, lut=lut, box=box, speckle=0.01, sharp=12.0)
        obs = scenes.make_obs(frame, agent, cam, goal_index=0)
        raised = False
        try:
            found_ref = ref.update(obs)
        except IndexError:
            raised = True
        if prob_rule:
            extra = dict(probs=_dev(frame["semantic_prob"][None], np.float32))
        else:
            extra = dict(semantic=_dev(frame["semantic"][None], np.int32),
                         semantic_unc=_dev(frame["semantic_unc"][None], np.float32),
                         semantic_max_prob=_dev(frame["semantic_max_prob"][None], np.float32))
        eng.step(_dev(frame["depth"][None], np.float32), agent[None], cam[None], [3],
                 semantic_gt=_dev(frame["semantic_gt"][None], np.int32), **extra)
        found, status, n_upd = eng.results()
        assert bool(status[0] & 1) == raised, f"frame {i}: numpy raised IndexError: {raised}, the engine reported: {bool(status[0] & 1)}"
        if raised:
            eng.close()
            return
        seen.append(bool(found_ref))
        assert bool(found[0]) == bool(found_ref), f"frame {i}: found ({rule})"
        sem = maps["semantic_map"].cpu().numpy().view(np.uint32)
        bad = sem != ref.semantic_map
        if prob_rule:
            parity.labels_agree(sem, ref.semantic_map, getattr(ref, name), f"frame {i}: semantic_map ({rule})")
        else:
            assert not bad.any(), f"frame {i}: semantic_map ({rule})"
    eng.close()
