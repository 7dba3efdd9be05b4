"""MapperGroup (N SemanticMapper-compatible views over one batched engine) against N independent
SemanticMappers: every map, found flag and ego map bit for bit, with views in different phases
(a round without a frame for one view, a reset in mid-run), from one thread and from one thread
per view; and the engine's skip mask on its own for every rule family."""
import threading

import numpy as np
import pytest
import torch

from tests import scenes

pytestmark = pytest.mark.gpu

BOUNDS = np.array([0.0, -0.2, 0.0, 4.5, 2.5, 4.5])
BOUNDS_B = np.array([0.0, -0.2, 0.0, 5.1, 2.5, 3.9])


def _kw(rule, c, filt=None):
    return dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
                prob_aggregate=rule, seg_model_n_classes=c, filter_goal_prob=filt)


def _frames(rule, c, n_views, n_frames, seed):
    """frames[t][i] = (obs, aux) of view i at its t-th call."""
    lut = scenes.small_lut(c)
    rng = np.random.default_rng(seed)
    out = []
    for t in range(n_frames):
        row = []
        for i in range(n_views):
            a, cm = scenes.make_pose(1.1 + 0.25 * i, 0.0, 1.0 + 0.3 * t, np.pi + 0.75 + 0.15 * i)
            fr = scenes.room_frame(rng, 60, 80, c, 1.0, cm, lut=lut, box=(2.0, 2.0, 2.9, 2.8, 0.5), speckle=0.01, sharp=8.0)
            aux = rng.random(12)
            row.append((scenes.make_obs(fr, a, cm), aux / aux.sum()))
        out.append(row)
    return out, lut


def _copy(obs):
    return {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in obs.items()}


def _same(res_a, res_b, tag):
    (ego_a, found_a), (ego_b, found_b) = res_a, res_b
    assert found_a == found_b, tag
    for x, y in zip(ego_a, ego_b):
        assert (x is None) == (y is None), tag
        if x is not None:
            assert np.array_equal(x, y), tag


def _same_maps(ma, mb, tag):
    from semantic_search_amd.mapper import _MAP_NAMES
    for nm in _MAP_NAMES:
        a, b = getattr(ma, nm, None), getattr(mb, nm, None)
        assert (a is None) == (b is None), (tag, nm)
        if a is not None:
            assert np.array_equal(a, b, equal_nan=True), (tag, nm)


SCHEDULE = [  # per round: which views have a frame; "reset" = that view resets before its frame
    {0: None, 1: None, 2: None},
    {0: None, 2: None},
    {0: None, 1: None, 2: None},
    {0: None, 1: "reset", 2: None},
    {1: None, 2: None},
    {0: None, 1: None, 2: None},
]


@pytest.mark.parametrize("rule,c,filt", [("height_map_avg_prob", 8, None), ("height_map_prob_hv", 8, None),
                                         ("height_map_goal_decay", 8, 0.35), ("height_map_log_odds", 8, None)])
def test_group_equals_independent_mappers(rule, c, filt):
    import semantic_search_amd as ssa
    frames, lut = _frames(rule, c, 3, len(SCHEDULE), seed=5)
    solo = [ssa.SemanticMapper(**_kw(rule, c, filt)) for _ in range(3)]
    cap = tuple(max(a, b) for a, b in zip(ssa.map_size(BOUNDS, 0.03), ssa.map_size(BOUNDS_B, 0.03)))
    group = ssa.MapperGroup(3, max_map_hw=cap, **_kw(rule, c, filt))
    for m in solo + group.mappers:
        m.map_seg_to_schema = lambda ids: lut[ids]
        if hasattr(m, "hv_min_views"):
            m.hv_min_views, m.hv_ratio = 1, 0.6
        if hasattr(m, "entropy_threshold"):
            m.entropy_threshold = 0.9
        m.reset(BOUNDS)
    calls = [0, 0, 0]
    for r, plan in enumerate(SCHEDULE):
        obs_list, aux_list, want = [None] * 3, [None] * 3, {}
        for i, what in plan.items():
            if what == "reset":
                solo[i].reset(BOUNDS_B)
                group.mappers[i].reset(BOUNDS_B)
            obs, aux = frames[calls[i]][i]
            calls[i] += 1
            want[i] = solo[i].update_map(_copy(obs), aux)
            obs_list[i], aux_list[i] = _copy(obs), aux
        got = group.update_maps(obs_list, aux_list)
        for i in range(3):
            if i in want:
                _same(got[i], want[i], (r, i))
                assert np.array_equal(obs_list[i]["depth"], np.where(np.isin(frames[calls[i] - 1][i][0]["depth"], (0.5, 5.0)), 0.0,
                                                                     frames[calls[i] - 1][i][0]["depth"]))
            else:
                assert got[i] is None
            _same_maps(group.mappers[i], solo[i], (r, i))
    group.close()


def test_group_views_from_one_thread_each():
    import semantic_search_amd as ssa
    rule, c = "height_map_avg_prob", 8
    frames, lut = _frames(rule, c, 3, 4, seed=9)
    solo = [ssa.SemanticMapper(**_kw(rule, c)) for _ in range(3)]
    group = ssa.MapperGroup(3, **_kw(rule, c))
    for m in solo + group.mappers:
        m.map_seg_to_schema = lambda ids: lut[ids]
        m.reset(BOUNDS)
    want = [[solo[i].update_map(_copy(frames[t][i][0]), frames[t][i][1]) for t in range(4)] for i in range(3)]
    got = [[None] * 4 for _ in range(3)]
    errors = []

    def run(i):
        try:
            for t in range(4):
                got[i][t] = group.mappers[i].update_map(_copy(frames[t][i][0]), frames[t][i][1])
        except Exception as exc:   # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=run, args=(i,)) for i in range(3)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    assert not errors, errors
    for i in range(3):
        for t in range(4):
            _same(got[i][t], want[i][t], (i, t))
        _same_maps(group.mappers[i], solo[i], i)
    # a view that calls alone is served after the timeout
    group.round_timeout_s = 0.2
    lone = group.mappers[1].update_map(_copy(frames[0][1][0]), frames[0][1][1])
    _same(lone, solo[1].update_map(_copy(frames[0][1][0]), frames[0][1][1]), "lone")
    group.close()


@pytest.mark.parametrize("rule", ["height_map_prob_hv", "height_map_hv", "height_map_goal_decay", "height_map_w_avg_prob",
                                  "height_map_argmax"])
def test_engine_skip_leaves_the_env_untouched(rule):
    """ssmap_frames.skip through the C ABI: a skipped env's maps are bit-identical before and after,
    its results read zero, and the other env steps as if alone."""
    import semantic_search_amd as ssa
    c, hw = 8, (48, 64)
    lut = scenes.small_lut(c)
    rng = np.random.default_rng(3)
    engs = [ssa.MapEngine(2, hw, max_map_hw=ssa.map_size(BOUNDS, 0.03), **_kw(rule, c)) for _ in range(2)]
    for eng in engs:
        eng.set_schema(lut)
        if rule.endswith("hv"):
            eng.params.hv_min_views, eng.params.hv_ratio = 1, 0.95   # rejections too
        for e in range(2):
            eng.reset_env(e, BOUNDS)
    prob_rule = "avg_prob" in rule

    def batch(t):
        fr, ag, cam = [], [], []
        for e in range(2):
            a, cm = scenes.make_pose(1.2 + 0.3 * e, 0.0, 1.0 + 0.3 * t, np.pi + 0.75)
            fr.append(scenes.room_frame(rng, hw[0], hw[1], c, 1.0, cm, lut=lut, box=(2.0, 2.0, 2.9, 2.8, 0.5), sharp=3.0))
            ag.append(a)
            cam.append(cm)
        dev = lambda key, dt: torch.from_numpy(np.stack([f[key] for f in fr]).astype(dt)).cuda()
        kw = dict(semantic_gt=dev("semantic_gt", np.int32))
        if prob_rule:
            kw["probs"] = dev("semantic_prob", np.float32)
        else:
            for key, dt in (("semantic", np.int32), ("semantic_unc", np.float32), ("semantic_max_prob", np.float32)):
                kw[key] = dev(key, dt)
        return dev("depth", np.float32), np.stack(ag), np.stack(cam), kw

    for t in range(5):
        depth, ag, cam, kw = batch(t)
        skipped = t in (1, 3)
        before = {nm: m.clone() for nm, m in engs[0].maps[1].items()}
        engs[0].step(depth.clone(), ag, cam, [3, 3], skip=[0, 1] if skipped else None,
                     **{k: v.clone() for k, v in kw.items()})
        found, status, n_upd = engs[0].results()
        if skipped:
            assert (found[1], status[1], n_upd[1]) == (False, 0, 0)
            for nm, m in engs[0].maps[1].items():
                assert torch.equal(m, before[nm]), nm
            # the reference engine sees env 1 frozen too: feed it env 1's frame only where env 0's ran
            engs[1].step(depth.clone(), ag, cam, [3, 3], skip=[0, 1], **{k: v.clone() for k, v in kw.items()})
        else:
            engs[1].step(depth.clone(), ag, cam, [3, 3], **{k: v.clone() for k, v in kw.items()})
        engs[1].results()
        for e in range(2):
            for nm, m in engs[0].maps[e].items():
                assert torch.equal(m, engs[1].maps[e][nm]), (t, e, nm)
        assert n_upd[0] > 50
    for eng in engs:
        eng.close()


def test_found_override_reads_the_views_own_env():
    """SpPolicyFoundOverride on a view of a MapperGroup asks the engine about that view's env: the
    views see different ground truth (only view 1's frames hold the goal), so env 0's answers on
    view 1 would flip its latches."""
    import semantic_search_amd as ssa
    rule, c = "height_map_argmax", 8
    frames, lut = _frames(rule, c, 3, 3, seed=21)
    goal_idx = 0
    for t in range(3):
        for i in range(3):
            obs = frames[t][i][0]
            obs["objectgoal"] = np.array([goal_idx])
            if i != 1:   # no goal in the ground truth of views 0 and 2
                obs["semantic_gt"] = np.where(obs["semantic_gt"] == 3, 2, obs["semantic_gt"])
            else:
                obs["semantic_gt"] = np.full_like(obs["semantic_gt"], 3)
    solo = [ssa.SemanticMapper(**_kw(rule, c)) for _ in range(3)]
    group = ssa.MapperGroup(3, **_kw(rule, c))
    for m in solo + group.mappers:
        m.map_seg_to_schema = lambda ids: lut[ids]
        m.reset(BOUNDS)
    sp_solo = [ssa.SpPolicyFoundOverride() for _ in range(3)]
    sp_view = [ssa.SpPolicyFoundOverride() for _ in range(3)]
    differs = False
    for t in range(3):
        want = [solo[i].update_map(_copy(frames[t][i][0]), None) for i in range(3)]
        got = group.update_maps([_copy(frames[t][i][0]) for i in range(3)])
        for i in range(3):
            # the found flag as the caller would pass it on; every branch is exercised by forcing it
            for forced in (want[i][1], True):
                a = sp_solo[i](solo[i], forced, 0.2, 1.0)
                b = sp_view[i](group.mappers[i], forced, 0.2, 1.0)
                assert (a, sp_solo[i].sp_fp_found, sp_solo[i].sp_fn_found) == \
                       (b, sp_view[i].sp_fp_found, sp_view[i].sp_fn_found), (t, i, forced)
        in_disk, anywhere = group._engine.gt_goal_query()
        differs |= bool(anywhere[1]) and not bool(anywhere[0])
    assert differs, "the views' ground truth should differ for this test to mean anything"
    group.close()


def test_goal_query_of_an_env_without_a_frame_since_bind_is_empty():
    """An env that every step since its ssmap_bind_env skipped has no robot cell and no goal: both
    answers of ssmap_gt_goal_query are 0, also after a reset in mid-run (the previous episode's
    robot cell and goal must not be asked about)."""
    import semantic_search_amd as ssa
    rule, c, hw = "height_map_argmax", 8, (48, 64)
    lut = scenes.small_lut(c)
    rng = np.random.default_rng(4)
    eng = ssa.MapEngine(2, hw, max_map_hw=ssa.map_size(BOUNDS, 0.03), **_kw(rule, c))
    eng.set_schema(lut)
    for e in range(2):
        eng.reset_env(e, BOUNDS)

    def step(skip):
        fr, ag, cam = [], [], []
        for e in range(2):
            a, cm = scenes.make_pose(1.2 + 0.3 * e, 0.0, 1.3, np.pi + 0.75)
            f = scenes.room_frame(rng, hw[0], hw[1], c, 1.0, cm, lut=lut)
            f["semantic_gt"] = np.full_like(f["semantic_gt"], 3)
            fr.append(f); ag.append(a); cam.append(cm)
        dev = lambda key, dt: torch.from_numpy(np.stack([f[key] for f in fr]).astype(dt)).cuda()
        eng.step(dev("depth", np.float32), np.stack(ag), np.stack(cam), [3, 3], skip=skip,
                 semantic_gt=dev("semantic_gt", np.int32), semantic=dev("semantic", np.int32),
                 semantic_unc=dev("semantic_unc", np.float32), semantic_max_prob=dev("semantic_max_prob", np.float32))
        eng.results()

    step([0, 1])                       # env 1 has never had a frame
    in_disk, anywhere = eng.gt_goal_query()
    assert bool(anywhere[0]) and not in_disk[1] and not anywhere[1]
    step(None)                         # now it has: its gt_map holds the goal
    in_disk, anywhere = eng.gt_goal_query()
    assert bool(anywhere[1])
    eng.reset_env(1, BOUNDS)           # new episode, then skipped again
    step([0, 1])
    in_disk, anywhere = eng.gt_goal_query()
    assert not in_disk[1] and not anywhere[1] and bool(anywhere[0])
    eng.close()


def test_skipped_envs_are_not_validated():
    """A skipped env's slices are neither read nor validated: a goal id outside [3, 8] there (what
    an idle view leaves behind) does not fail a goal-statistics step of the other env."""
    import semantic_search_amd as ssa
    rule, c, hw = "height_map_argmax", 8, (48, 64)
    lut = scenes.small_lut(c)
    rng = np.random.default_rng(8)
    eng = ssa.MapEngine(2, hw, max_map_hw=ssa.map_size(BOUNDS, 0.03), collect_nb_data=True, **_kw(rule, c))
    eng.set_schema(lut)
    for e in range(2):
        eng.reset_env(e, BOUNDS)
    a, cm = scenes.make_pose(1.2, 0.0, 1.3, np.pi + 0.75)
    f = scenes.room_frame(rng, hw[0], hw[1], c, 1.0, cm, lut=lut)
    dev = lambda key, dt: torch.from_numpy(np.stack([f[key], f[key]]).astype(dt)).cuda()
    kw = dict(semantic_gt=dev("semantic_gt", np.int32), semantic=dev("semantic", np.int32),
              semantic_unc=dev("semantic_unc", np.float32), semantic_max_prob=dev("semantic_max_prob", np.float32),
              probs=dev("semantic_prob", np.float32))
    eng.step(dev("depth", np.float32), np.stack([a, a]), np.stack([cm, cm]), [3, 0], skip=[0, 1], **kw)
    found, status, n_upd = eng.results()
    assert n_upd[0] > 50 and n_upd[1] == 0
    with pytest.raises(Exception, match="goal id outside"):
        eng.step(dev("depth", np.float32), np.stack([a, a]), np.stack([cm, cm]), [3, 0], **kw)
    eng.close()
