"""HIP engine against the CPU oracle on seeded inputs, through the C ABI.

Bars: update set (cell indices), integer maps and found flags bit-exact; float32 state within
rtol 1e-5 / atol 1e-6.  Where a label depends on an argmax over floating-point values that the
two implementations round differently, a mismatch is accepted only if the two leading values
are within 1e-5 relative of each other, and the number of such cells is reported.
"""
import numpy as np
import pytest
import torch

from oracle import mapper_oracle as mo
from oracle import stage1_oracle as s1
from tests import parity, scenes

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-6
# Worst element-wise relative error of a log-odds entry (|x| >= 1e-3) on the full-size frames: the
# hardware logarithm is good to an ulp or two of log2(q), i.e. an absolute 2e-7 on a term of
# magnitude 1-10; an entry that cancels to 1e-3 keeps that absolute error.
ELEMENTWISE_LOG_ODDS_BOUND = 1e-3
BOUNDS_480 = np.array([0.0, 0.0, 0.0, 14.39, 3.0, 14.39])


def _close(got, want, name, rtol=RTOL, atol=ATOL):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"{name}: NaN pattern"
    assert np.array_equal(np.isinf(got), np.isinf(want)), f"{name}: inf pattern"
    fin = np.isfinite(want)
    err = np.abs(got[fin] - want[fin])
    tol = atol + rtol * np.abs(want[fin])
    assert (err <= tol).all(), f"{name}: max err {err.max():.3e} (ratio {(err / tol).max():.2f})"


def _to_dev(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


def _top2_margin(vectors):
    srt = np.sort(vectors.astype(np.float64), axis=-1)
    return (srt[..., -1] - srt[..., -2]) / np.maximum(np.abs(srt[..., -1]), 1e-30)


@pytest.mark.parametrize("n_classes,temperature,generic", [(40, 3.2, False), (150, 2.5, False), (12, 1.0, False), (53, 2.0, False),
                                                           (8, 1.0, False), (21, 3.2, False), (100, 2.5, False), (155, 1.0, False),
                                                           (12, 1.0, True), (40, 3.2, True)])
def test_stage1_matches_oracle(n_classes, temperature, generic, monkeypatch):
    """Class counts with a kernel of their own, counts that run the next size up (12, 21, 100, 155), and
    the generic LDS-column kernel (SSMAP_GENERIC=1 when the engine is created)."""
    import semantic_search_amd as ssa
    monkeypatch.setenv("SSMAP_GENERIC", "1" if generic else "0")
    h, w = 120, 160
    rng = np.random.default_rng(n_classes)
    logits = (3.0 * rng.standard_normal((2, n_classes, h, w))).astype(np.float32)
    lut = scenes.schema_lut(n_classes) if n_classes >= 16 else scenes.small_lut(n_classes)
    eng = ssa.MapEngine(2, (h, w), depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
                        prob_aggregate="height_map_argmax", seg_model_n_classes=n_classes,
                        max_map_hw=(100, 100))
    eng.set_schema(lut)
    out = eng.stage1(_to_dev(logits, np.float32), temperature, want_probs=True)
    torch.cuda.synchronize()
    for e in range(2):
        want = s1.predict_tail(logits[e], temperature, lut)
        probs = out["semantic_prob"][e].cpu().numpy()
        _close(probs, want["semantic_prob"], "semantic_prob", atol=1e-9)
        _close(out["semantic_unc"][e].cpu().numpy(), want["semantic_unc"], "semantic_unc")
        sem = out["semantic"][e].cpu().numpy()
        bad = sem != want["semantic"]
        if bad.any():   # only genuine argmax ties may differ
            margins = _top2_margin(np.moveaxis(want["semantic_prob"], 0, -1)[bad])
            assert (margins < 1e-5).all()
        ok = ~bad
        _close(out["semantic_max_prob"][e].cpu().numpy()[ok], want["semantic_max_prob"][ok], "semantic_max_prob")
    eng.close()


@pytest.mark.parametrize("n_classes", [40, 13, 150])
def test_stage1_edge_logits(n_classes):
    """Exact ties (argmax takes the lowest class), one class dominating until the softmax saturates,
    classes masked with -inf, and logits of large magnitude: labels bit-exact, floats within rtol
    1e-5.  13 classes take the generic kernel, 40 the register kernel."""
    import semantic_search_amd as ssa
    h, w = 24, 32
    rng = np.random.default_rng(7)
    logits = (2.0 * rng.standard_normal((1, n_classes, h, w))).astype(np.float32)
    logits[0, :, 0, :] = 1.25                                  # every class equal
    logits[0, :, 1, :] = 0.0
    logits[0, 3, 1, :] = 0.5
    logits[0, 7, 1, :] = 0.5                                   # two leaders tie exactly: class 3 wins
    logits[0, :, 2, :] = -3.0
    logits[0, 5, 2, :] = 400.0                                 # saturated: p = 1, entropy = 0
    logits[0, 1::2, 3, :] = -np.inf                            # masked classes
    logits[0, :, 4, :] *= 1.0e4                                # large magnitudes, both signs (compared at T = 1)
    logits[0, :, 6, :] *= 30.0
    logits[0, :, 5, :] = -1.0e30
    logits[0, 2, 5, :] = -1.0e30 + 1.0e24
    lut = scenes.schema_lut(n_classes) if n_classes >= 37 else scenes.small_lut(n_classes)
    eng = ssa.MapEngine(1, (h, w), depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
                        prob_aggregate="height_map_argmax", seg_model_n_classes=n_classes, max_map_hw=(100, 100))
    eng.set_schema(lut)
    for temperature in (1.0, 3.2):
        if temperature != 1.0:
            # x / T (the oracle, torch on the CPU) and x * (1 / T) (torch on a GPU, and the engine)
            # differ by an ulp of the scaled logit now and then, which at 1e4 is 1e-3: that row is
            # only comparable where the division is exact
            logits[0, :, 4, :] = logits[0, :, 6, :]
        out = eng.stage1(_to_dev(logits, np.float32), temperature, want_probs=True)
        torch.cuda.synchronize()
        want = s1.predict_tail(logits[0], temperature, lut)
        probs = out["semantic_prob"][0].cpu().numpy()
        assert np.isfinite(probs).all() and np.isfinite(want["semantic_prob"]).all()
        _close(probs, want["semantic_prob"], "semantic_prob", atol=1e-9)
        _close(out["semantic_unc"][0].cpu().numpy(), want["semantic_unc"], "semantic_unc", atol=2e-6)
        sem = out["semantic"][0].cpu().numpy()
        rows = slice(0, 7)                                      # the constructed rows have no near-ties
        assert np.array_equal(sem[rows], want["semantic"][rows])
        assert (sem[1] == lut[3]).all() and (sem[2] == lut[5]).all()
        _close(out["semantic_max_prob"][0].cpu().numpy()[rows], want["semantic_max_prob"][rows], "semantic_max_prob")
    eng.close()


@pytest.mark.parametrize("n_classes", [40, 13, 150])
def test_stage1_goal_confidence_follows_the_pool_lists(n_classes):
    """semantic_max_prob of a goal pixel is the sum over the goal's listed classes in list order
    (infer.py:109-118, 318-327), whatever the class table maps to that goal: a list that is not the
    table's inverse (class 9 maps to chair but is not pooled), one in non-ascending order with four
    members, and a goal whose list is empty (its pixels get 0).  13 classes take the generic
    kernel, 40 and 150 the register kernels; both must give the same bits as each other's sums, so
    the comparison with the oracle is at 1e-6 relative."""
    import semantic_search_amd as ssa
    h, w = 96, 128
    rng = np.random.default_rng(100 + n_classes)
    logits = (1.5 * rng.standard_normal((1, n_classes, h, w))).astype(np.float32)
    lut = np.full(n_classes, 2, dtype=np.int64)
    lut[[3, 9]] = 3
    lut[[12, 5, 11, 7]] = 7
    lut[[4]] = 4
    lut[[6]] = 8
    lut[1] = 1
    pools = {3: [3], 7: [12, 5, 11, 7], 4: [4, 2], 8: []}
    eng = ssa.MapEngine(1, (h, w), depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
                        prob_aggregate="height_map_argmax", seg_model_n_classes=n_classes, max_map_hw=(100, 100))
    eng.set_schema(lut, pools)
    out = eng.stage1(_to_dev(logits, np.float32), 1.0, want_probs=True)
    torch.cuda.synchronize()
    want = s1.predict_tail(logits[0], 1.0, lut, pool_groups=pools)
    sem = out["semantic"][0].cpu().numpy()
    ok = sem == want["semantic"]
    assert ok.mean() > 0.999
    for g in (3, 4, 7, 8):
        assert (want["semantic"] == g).sum() > 20, g
    got = out["semantic_max_prob"][0].cpu().numpy()
    assert (got[ok & (sem == 8)] == 0).all()
    _close(got[ok], want["semantic_max_prob"][ok], "semantic_max_prob", rtol=1e-6, atol=1e-9)
    # exactly the float32 sum of the engine's own probabilities in list order
    probs = out["semantic_prob"][0].cpu().numpy()
    acc = probs[12] + probs[5]
    acc = acc + probs[11]
    acc = acc + probs[7]
    assert np.array_equal(got[sem == 7], acc[sem == 7])
    eng.close()


def _run_pair(rule, n_classes, temperature, n_frames, hw, from_logits, attrs=None, filter_goal_prob=None,
              seed=0, bounds=BOUNDS_480, center=7.2):
    """Drives the engine (E=1, device-resident inputs) and the oracle over the same frames."""
    import semantic_search_amd as ssa
    h, w = hw
    lut = scenes.schema_lut(n_classes) if n_classes >= 16 else scenes.small_lut(n_classes)
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate=rule, seg_model_n_classes=n_classes, filter_goal_prob=filter_goal_prob)
    ref = mo.OracleMapper(schema_lut=lut, path_overlay=False, **kw)
    eng = ssa.MapEngine(1, hw, max_map_hw=ssa.map_size(bounds, 0.03), **kw)
    eng.set_schema(lut)
    for k, v in (attrs or {}).items():
        setattr(ref, k, v)
        setattr(eng.params, k, v)
    ref.reset(bounds)
    maps = eng.reset_env(0, bounds)
    rng = np.random.default_rng(seed)
    prng = np.random.default_rng(seed + 100)
    log = []
    for i in range(n_frames):
        frame = scenes.uniform_frame(rng, h, w, n_classes, temperature, lut=lut, with_probs=False)
        agent, cam = scenes.make_pose(center, 0.0, center, prng.uniform(0, 2 * np.pi))
        st = s1.predict_tail(frame["logits"], temperature, lut)
        obs = {"depth": frame["depth"].copy(), "semantic_gt": frame["semantic_gt"], "objectgoal": np.array([0]),
               "agent_position_and_rotation": agent, "camera_position_and_rotation": cam,
               "semantic": st["semantic"], "semantic_prob": st["semantic_prob"],
               "semantic_unc": st["semantic_unc"], "semantic_max_prob": st["semantic_max_prob"]}
        found_ref = ref.update(obs)
        depth = _to_dev(frame["depth"][None], np.float32)
        gt = _to_dev(frame["semantic_gt"][None], np.int32)
        sem = torch.empty((1, h, w), dtype=torch.int32, device="cuda")
        unc = torch.empty((1, h, w), dtype=torch.float32, device="cuda")
        maxp = torch.empty((1, h, w), dtype=torch.float32, device="cuda")
        if from_logits:
            eng.step(depth, agent[None], cam[None], [3], logits=_to_dev(frame["logits"][None], np.float32),
                     temperature=temperature, semantic=sem, semantic_unc=unc, semantic_max_prob=maxp,
                     semantic_gt=gt)
        else:
            eng.step(depth, agent[None], cam[None], [3], probs=_to_dev(st["semantic_prob"][None], np.float32),
                     semantic=_to_dev(st["semantic"][None], np.int32),
                     semantic_unc=_to_dev(st["semantic_unc"][None], np.float32),
                     semantic_max_prob=_to_dev(st["semantic_max_prob"][None], np.float32), semantic_gt=gt)
        found, status, n_upd = eng.results()
        assert status[0] == 0
        pix, _ = eng.update_list(0, int(n_upd[0]))
        assert np.array_equal(pix, ref.last_update_pixels), f"frame {i}: update set"
        assert np.array_equal(depth[0].cpu().numpy(), obs["depth"]), "depth sentinels"
        log.append((bool(found[0]), bool(found_ref)))
    return eng, maps, ref, log


@pytest.mark.parametrize("rule,from_logits", [
    ("height_map_log_odds", True), ("height_map_log_odds", False), ("height_map_avg_prob", True), ("height_map_avg_prob", False),
    ("height_map_w_avg_prob", True), ("height_map_w2_avg_prob", False)])
def test_full_size_probability_rules(rule, from_logits):
    """640x480x40 benchmark-recipe frames, two frames, fused state vs the oracle."""
    eng, maps, ref, log = _run_pair(rule, 40, 3.2, 2, (480, 640), from_logits,
                                    attrs=dict(entropy_threshold=0.43759))
    name = "log_odds_seg_map" if "log_odds" in rule else "avg_prob_seg_map"
    got = maps[name].cpu().numpy()
    want = getattr(ref, name)
    if "log_odds" in rule:
        parity.close_rows(got, want, name)          # relative to the cell's largest |log-odds|
        # ... and element by element: an entry is a sum of a few log(p_c / p_0) that can cancel, so
        # its own magnitude is a harder yardstick than its cell's; the worst entry is reported
        worst, above, n = parity.elementwise_relative(got, want, floor=1e-3)
        print(f"\n{name}: element-wise relative error over {n} entries with |x| >= 1e-3: worst {worst:.2e}, "
              f"{100 * above:.4f} % above 1e-5")
        assert worst <= ELEMENTWISE_LOG_ODDS_BOUND, worst
    else:
        _close(got, want, name)
    if rule == "height_map_avg_prob" and not from_logits:
        assert np.array_equal(got, want), "unit weights, same inputs: the running average is exact"
    assert np.array_equal(maps["height_map"].cpu().numpy(), ref.height_map)
    assert np.array_equal(maps["gt_map"].cpu().numpy().view(np.uint32), ref.gt_map)
    sem = maps["semantic_map"].cpu().numpy().view(np.uint32)
    flips = parity.labels_agree(sem, ref.semantic_map, want)
    print(f"\n{rule}: {flips} of {int((ref.semantic_map != 0).sum())} labelled cells differ from the oracle's "
          "(top-2 within 1e-5, no goal id)")
    if "avg" in rule:
        _close(maps["unc_map"].cpu().numpy(), ref.unc_map, "unc_map")
    if not flips:
        assert [a for a, _ in log] == [b for _, b in log]
    eng.close()


def test_frame_above_packed_record_limit():
    """768x704 = 540 672 pixels: above 2^19, so the binned records carry slot and pixel in
    separate words.  Update set, height and ground-truth maps exact, fused map within tolerance."""
    eng, maps, ref, log = _run_pair("height_map_avg_prob", 8, 1.0, 2, (704, 768), False,
                                    attrs=dict(entropy_threshold=0.43759))
    _close(maps["avg_prob_seg_map"].cpu().numpy(), ref.avg_prob_seg_map, "avg_prob_seg_map")
    assert np.array_equal(maps["height_map"].cpu().numpy(), ref.height_map)
    assert np.array_equal(maps["gt_map"].cpu().numpy().view(np.uint32), ref.gt_map)
    eng.close()


def test_far_depths_take_the_unbinned_winner_path():
    """Depths up to 30 m on a 960 x 960 map with depth_range up to 40 m: the predicted cell box is
    far larger than the winner tables hold, so every table block scans the pixels instead."""
    import semantic_search_amd as ssa
    h, w, c = 120, 160, 8
    lut = scenes.small_lut(c)
    bounds = np.array([0.0, 0.0, 0.0, 28.79, 3.0, 28.79])
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 40.0), probablistic_mapping=True,
              prob_aggregate="height_map_avg_prob", seg_model_n_classes=c)
    ref = mo.OracleMapper(schema_lut=lut, path_overlay=False, **kw)
    eng = ssa.MapEngine(1, (h, w), max_map_hw=ssa.map_size(bounds, 0.03), **kw)
    eng.set_schema(lut)
    ref.reset(bounds)
    maps = eng.reset_env(0, bounds)
    rng, prng = np.random.default_rng(21), np.random.default_rng(22)
    for i in range(2):
        frame = scenes.uniform_frame(rng, h, w, c, 1.0, depth_range=(0.5, 40.0), lut=lut)
        frame["depth"] = np.where(frame["depth"] == np.float32(0.5), frame["depth"],
                                  rng.uniform(1.0, 30.0, (h, w)).astype(np.float32))
        agent, cam = scenes.make_pose(14.0, 0.0, 14.0, prng.uniform(0, 2 * np.pi))
        obs = scenes.make_obs(frame, agent, cam, goal_index=0)
        obs["depth"] = frame["depth"].copy()
        ref.update(obs)
        eng.step(_to_dev(frame["depth"][None], np.float32), agent[None], cam[None], [3],
                 probs=_to_dev(frame["semantic_prob"][None], np.float32),
                 semantic_gt=_to_dev(frame["semantic_gt"][None], np.int32))
        found, status, n_upd = eng.results()
        assert status[0] == 0
        pix, _ = eng.update_list(0, int(n_upd[0]))
        assert np.array_equal(np.sort(pix), np.sort(ref.last_update_pixels)), f"frame {i}: update set"
    rows = np.nonzero(np.isfinite(ref.height_map).any(axis=1))[0]
    assert rows.max() - rows.min() > 500, "the scene must spread over more rows than the tables can bin"
    _close(maps["avg_prob_seg_map"].cpu().numpy(), ref.avg_prob_seg_map, "avg_prob_seg_map")
    assert np.array_equal(maps["height_map"].cpu().numpy(), ref.height_map)
    assert np.array_equal(maps["gt_map"].cpu().numpy().view(np.uint32), ref.gt_map)
    eng.close()


def _drive(eng, ref, maps, frames_poses, h, w):
    for i, (frame, agent, cam) in enumerate(frames_poses):
        obs = scenes.make_obs(frame, agent, cam, goal_index=0)
        obs["depth"] = frame["depth"].copy()
        ref.update(obs)
        eng.step(_to_dev(frame["depth"][None], np.float32), agent[None], cam[None], [3],
                 probs=_to_dev(frame["semantic_prob"][None], np.float32),
                 semantic_gt=_to_dev(frame["semantic_gt"][None], np.int32))
        found, status, n_upd = eng.results()
        assert status[0] == 0
        pix, _ = eng.update_list(0, int(n_upd[0]))
        assert np.array_equal(np.sort(pix), np.sort(ref.last_update_pixels)), f"frame {i}: update set"
    _close(maps["avg_prob_seg_map"].cpu().numpy(), ref.avg_prob_seg_map, "avg_prob_seg_map")
    assert np.array_equal(maps["height_map"].cpu().numpy(), ref.height_map)
    assert np.array_equal(maps["gt_map"].cpu().numpy().view(np.uint32), ref.gt_map)


def test_depths_beyond_depth_max_leave_the_predicted_box():
    """depth_range (0.5, 5.0) but a few pixels at 6 to 9 m (only the exact maximum is a sentinel,
    mapper.py:279-283): their cells lie outside the cell box the host predicts from the view
    pyramid up to depth_max, so the frame takes the scanning winner path.  The second frame stays
    inside the range and is binned again."""
    import semantic_search_amd as ssa
    h, w, c = 120, 160, 8
    lut = scenes.small_lut(c)
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate="height_map_avg_prob", seg_model_n_classes=c)
    ref = mo.OracleMapper(schema_lut=lut, path_overlay=False, **kw)
    eng = ssa.MapEngine(1, (h, w), max_map_hw=ssa.map_size(BOUNDS_480, 0.03), **kw)
    eng.set_schema(lut)
    ref.reset(BOUNDS_480)
    maps = eng.reset_env(0, BOUNDS_480)
    rng, prng = np.random.default_rng(31), np.random.default_rng(32)
    seq = []
    for i in range(2):
        frame = scenes.uniform_frame(rng, h, w, c, 1.0, lut=lut)
        if i == 0:
            far = rng.random((h, w)) < 0.01
            frame["depth"] = np.where(far, rng.uniform(6.0, 9.0, (h, w)).astype(np.float32), frame["depth"])
        agent, cam = scenes.make_pose(7.2, 0.0, 7.2, prng.uniform(0, 2 * np.pi))
        seq.append((frame, agent, cam))
    _drive(eng, ref, maps, seq, h, w)
    far_rows = np.nonzero(np.isfinite(ref.height_map).any(axis=1))[0]
    assert far_rows.max() - far_rows.min() > 2 * int(5.0 / 0.03) * 0.6, "some cells must lie beyond the 5 m reach"
    eng.close()


def test_frame_of_more_than_a_million_pixels():
    """1056 x 1024 pixels = 1056 geometry blocks, more than a winner block keeps books for: the
    frame takes the scanning winner path whatever its cell box."""
    import semantic_search_amd as ssa
    h, w, c = 1056, 1024, 5
    lut = scenes.small_lut(c)
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate="height_map_avg_prob", seg_model_n_classes=c)
    ref = mo.OracleMapper(schema_lut=lut, path_overlay=False, **kw)
    eng = ssa.MapEngine(1, (h, w), max_map_hw=ssa.map_size(BOUNDS_480, 0.03), **kw)
    eng.set_schema(lut)
    ref.reset(BOUNDS_480)
    maps = eng.reset_env(0, BOUNDS_480)
    rng, prng = np.random.default_rng(41), np.random.default_rng(42)
    frame = scenes.uniform_frame(rng, h, w, c, 1.0, lut=lut)
    agent, cam = scenes.make_pose(7.2, 0.0, 7.2, prng.uniform(0, 2 * np.pi))
    _drive(eng, ref, maps, [(frame, agent, cam)], h, w)
    eng.close()


def test_stubborn_rule_from_logits_equals_stubborn_from_stage1_outputs():
    """height_map_stubborn, C = 40, two engines over the same frames: one is fed logits (stage 1
    runs inside the step and leaves the probabilities for the goal statistics), the other the
    stage-1 outputs as tensors.  Same maps, same statistics, and the model vetoes some goal cells."""
    import semantic_search_amd as ssa
    from tests import golden_util as gu
    h, w, c, temp = 120, 160, 40, 3.2
    lut = scenes.schema_lut(c)
    model = gu.GoldenCase("stubborn_nb").mapper_kwargs()["nb_models"][3]
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate="height_map_stubborn", seg_model_n_classes=c)
    engines = []
    for _ in range(2):
        eng = ssa.MapEngine(1, (h, w), max_map_hw=ssa.map_size(BOUNDS_480, 0.03), **kw)
        eng.set_schema(lut)
        eng.set_nb_model(3, model)
        engines.append((eng, eng.reset_env(0, BOUNDS_480)))
    rng, prng = np.random.default_rng(51), np.random.default_rng(52)
    for i in range(3):
        frame = scenes.uniform_frame(rng, h, w, c, temp, lut=lut, with_probs=False)
        agent, cam = scenes.make_pose(7.2, 0.0, 7.2, prng.uniform(0, 2 * np.pi))
        logits = _to_dev(frame["logits"][None], np.float32)
        gt = _to_dev(frame["semantic_gt"][None], np.int32)
        (ea, _), (eb, _) = engines
        sem = torch.empty((1, h, w), dtype=torch.int32, device="cuda")
        unc = torch.empty((1, h, w), dtype=torch.float32, device="cuda")
        maxp = torch.empty((1, h, w), dtype=torch.float32, device="cuda")
        ea.step(_to_dev(frame["depth"][None], np.float32), agent[None], cam[None], [3], logits=logits,
                temperature=temp, semantic=sem, semantic_unc=unc, semantic_max_prob=maxp, semantic_gt=gt)
        st = eb.stage1(logits, temp, want_probs=True)
        eb.step(_to_dev(frame["depth"][None], np.float32), agent[None], cam[None], [3],
                probs=st["semantic_prob"], semantic=st["semantic"], semantic_unc=st["semantic_unc"],
                semantic_max_prob=st["semantic_max_prob"], semantic_gt=gt)
        assert not ea.results()[1].any() and not eb.results()[1].any()
    ma, mb = engines[0][1], engines[1][1]
    for name in ("semantic_map", "height_map", "gt_map", "unc_map", "total_view", "cumulative_conf",
                 "max_conf", "max_other_conf"):
        assert torch.equal(ma[name], mb[name]), name
    assert int(ma["total_view"].max()) >= 2
    for eng, _ in engines:
        eng.close()


def test_segformer_style_150_classes_w2():
    eng, maps, ref, log = _run_pair("height_map_w2_avg_prob", 150, 2.5, 2, (120, 160), True,
                                    attrs=dict(entropy_threshold=0.482), bounds=BOUNDS_480)
    _close(maps["avg_prob_seg_map"].cpu().numpy(), ref.avg_prob_seg_map, "avg_prob_seg_map")
    _close(maps["avg_prob_weight_map"].cpu().numpy(), ref.avg_prob_weight_map, "weight")
    _close(maps["unc_map"].cpu().numpy(), ref.unc_map, "unc_map")
    assert np.array_equal(maps["height_map"].cpu().numpy(), ref.height_map)
    eng.close()


@pytest.mark.parametrize("rule,c,from_logits", [("height_map_log_odds", 37, True), ("height_map_w2_avg_prob", 81, True),
                                                ("height_map_avg_prob", 53, False), ("height_map_w_avg_prob", 37, False),
                                                ("height_map_log_odds", 100, True), ("height_map_w2_avg_prob", 155, True),
                                                ("height_map_avg_prob", 21, False), ("height_map_log_odds", 21, True)])
def test_other_heads_of_the_segmentation_wrapper(rule, c, from_logits):
    """The class counts of the reference's other EMSANet heads (infer.py:37-95) and counts without a
    kernel of their own (21, 100, 155: the kernel of the next size up): the streaming kernel's own
    update against the oracle, fed logits or probabilities."""
    eng, maps, ref, log = _run_pair(rule, c, 3.2, 3, (120, 160), from_logits, attrs=dict(entropy_threshold=0.3),
                                    bounds=BOUNDS_480)
    if "log_odds" in rule:
        parity.close_rows(maps["log_odds_seg_map"].cpu().numpy(), ref.log_odds_seg_map, "log_odds_seg_map")
    else:
        _close(maps["avg_prob_seg_map"].cpu().numpy(), ref.avg_prob_seg_map, "avg_prob_seg_map")
        _close(maps["avg_prob_weight_map"].cpu().numpy(), ref.avg_prob_weight_map, "weight")
        _close(maps["unc_map"].cpu().numpy(), ref.unc_map, "unc_map")
    assert np.array_equal(maps["height_map"].cpu().numpy(), ref.height_map)
    assert np.array_equal(maps["gt_map"].cpu().numpy(), ref.gt_map)
    parity.labels_agree(maps["semantic_map"].cpu().numpy(), ref.semantic_map,
                        ref.log_odds_seg_map if "log_odds" in rule else ref.avg_prob_seg_map, "semantic_map")
    eng.close()


def test_segformer_style_150_classes_w2_full_size_batched():
    """BASELINE configs[2] at full frame size: height_map_w2_avg_prob, 150 classes, T = 2.5, logits
    in, 640x480 frames, four envs in one call, two frames each, every env against its own oracle."""
    import semantic_search_amd as ssa
    h, w, c, t, E = 480, 640, 150, 2.5, 4
    lut = scenes.schema_lut(c)
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate="height_map_w2_avg_prob", seg_model_n_classes=c)
    eng = ssa.MapEngine(E, (h, w), max_map_hw=ssa.map_size(BOUNDS_480, 0.03), **kw)
    eng.set_schema(lut)
    eng.params.entropy_threshold = 0.482
    refs = []
    for e in range(E):
        eng.reset_env(e, BOUNDS_480)
        r = mo.OracleMapper(schema_lut=lut, path_overlay=False, **kw)
        r.entropy_threshold = 0.482
        r.reset(BOUNDS_480)
        refs.append(r)
    rng, prng = np.random.default_rng(71), np.random.default_rng(72)
    for i in range(2):
        frames = [scenes.uniform_frame(rng, h, w, c, t, lut=lut, with_probs=False) for _ in range(E)]
        poses = [scenes.make_pose(7.2 + 0.1 * e, 0.0, 7.2, prng.uniform(0, 2 * np.pi)) for e in range(E)]
        sem = torch.empty((E, h, w), dtype=torch.int32, device="cuda")
        unc = torch.empty((E, h, w), dtype=torch.float32, device="cuda")
        maxp = torch.empty((E, h, w), dtype=torch.float32, device="cuda")
        eng.step(_to_dev(np.stack([f["depth"] for f in frames]), np.float32),
                 np.stack([p[0] for p in poses]), np.stack([p[1] for p in poses]), [3] * E,
                 logits=_to_dev(np.stack([f["logits"] for f in frames]), np.float32), temperature=t,
                 semantic=sem, semantic_unc=unc, semantic_max_prob=maxp,
                 semantic_gt=_to_dev(np.stack([f["semantic_gt"] for f in frames]), np.int32))
        found, status, n_upd = eng.results()
        assert not status.any()
        for e in range(E):
            st = s1.predict_tail(frames[e]["logits"], t, lut)
            obs = {"depth": frames[e]["depth"].copy(), "semantic_gt": frames[e]["semantic_gt"],
                   "objectgoal": np.array([0]), "agent_position_and_rotation": poses[e][0],
                   "camera_position_and_rotation": poses[e][1], "semantic_prob": st["semantic_prob"]}
            with np.errstate(all="ignore"):
                refs[e].update(obs)
            pix, _ = eng.update_list(e, int(n_upd[e]))
            assert np.array_equal(pix, refs[e].last_update_pixels), (i, e)
    for e in range(E):
        maps, ref = eng.maps[e], refs[e]
        _close(maps["avg_prob_seg_map"].cpu().numpy(), ref.avg_prob_seg_map, f"avg_prob_seg_map[{e}]")
        _close(maps["avg_prob_weight_map"].cpu().numpy(), ref.avg_prob_weight_map, f"weight[{e}]")
        _close(maps["unc_map"].cpu().numpy(), ref.unc_map, f"unc_map[{e}]")
        assert np.array_equal(maps["height_map"].cpu().numpy(), ref.height_map)
        parity.labels_agree(maps["semantic_map"].cpu().numpy().view(np.uint32), ref.semantic_map, ref.avg_prob_seg_map)
    eng.close()


def test_prob_hv_on_the_960_map():
    """BASELINE configs[4]'s map: height_map_prob_hv on a 960x960 map (bounds 28.79 m), 640x480x40
    logits in, two envs, three frames.  The oracle is fed the engine's own stage-1 planes, so every
    map must be equal bit for bit."""
    import semantic_search_amd as ssa
    h, w, c, t, E = 480, 640, 40, 3.2, 2
    bounds = np.array([0.0, 0.0, 0.0, 28.79, 3.0, 28.79])
    lut = scenes.schema_lut(c)
    attrs = dict(hv_ratio=0.887, hv_min_views=5, hv_limit_obj_view_dist=4.13219)
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate="height_map_prob_hv", seg_model_n_classes=c)
    eng = ssa.MapEngine(E, (h, w), max_map_hw=ssa.map_size(bounds, 0.03), **kw)
    assert eng.max_map_hw == (960, 960)
    eng.set_schema(lut)
    refs = []
    for e in range(E):
        eng.reset_env(e, bounds)
        r = mo.OracleMapper(schema_lut=lut, path_overlay=False, **kw)
        r.reset(bounds)
        refs.append(r)
    for k, v in attrs.items():
        setattr(eng.params, k, v)
        for r in refs:
            setattr(r, k, v)
    rng, prng = np.random.default_rng(81), np.random.default_rng(82)
    for i in range(3):
        frames = [scenes.uniform_frame(rng, h, w, c, t, lut=lut, with_probs=False) for _ in range(E)]
        poses = [scenes.make_pose(14.4, 0.0, 14.4 + 0.2 * e, prng.uniform(0, 2 * np.pi)) for e in range(E)]
        sem = torch.empty((E, h, w), dtype=torch.int32, device="cuda")
        unc = torch.empty((E, h, w), dtype=torch.float32, device="cuda")
        maxp = torch.empty((E, h, w), dtype=torch.float32, device="cuda")
        eng.step(_to_dev(np.stack([f["depth"] for f in frames]), np.float32),
                 np.stack([p[0] for p in poses]), np.stack([p[1] for p in poses]), [3] * E,
                 logits=_to_dev(np.stack([f["logits"] for f in frames]), np.float32), temperature=t,
                 semantic=sem, semantic_unc=unc, semantic_max_prob=maxp,
                 semantic_gt=_to_dev(np.stack([f["semantic_gt"] for f in frames]), np.int32))
        found, status, n_upd = eng.results()
        assert not status.any()
        for e in range(E):
            obs = {"depth": frames[e]["depth"].copy(), "semantic_gt": frames[e]["semantic_gt"],
                   "objectgoal": np.array([0]), "agent_position_and_rotation": poses[e][0],
                   "camera_position_and_rotation": poses[e][1], "semantic": sem[e].cpu().numpy(),
                   "semantic_unc": unc[e].cpu().numpy(), "semantic_max_prob": maxp[e].cpu().numpy()}
            found_ref = refs[e].update(obs)
            assert bool(found[e]) == bool(found_ref), (i, e)
            pix, _ = eng.update_list(e, int(n_upd[e]))
            assert np.array_equal(pix, refs[e].last_update_pixels), (i, e)
    for e in range(E):
        for nm in ("semantic_map", "height_map", "gt_map", "background_map", "hits", "views", "unc_map"):
            got = eng.maps[e][nm].cpu().numpy()
            if got.dtype == np.int32:
                got = got.view(np.uint32)
            assert np.array_equal(got, getattr(refs[e], nm)), (e, nm)
    eng.close()


@pytest.mark.parametrize("rule,attrs,filt", [
    ("height_map_goal_decay", dict(goal_decay=0.8931, goal_mark=1.5408), 0.38158),
    ("height_map_prob_hv", dict(hv_ratio=0.887, hv_min_views=5, hv_limit_obj_view_dist=4.13219), None),
    ("height_map_hv", dict(hv_ratio=0.5, hv_min_views=1, hv_limit_obj_view_dist=0.0), 0.3),
    ("height_map_argmax", {}, 0.53446)])
def test_full_size_label_rules_from_logits(rule, attrs, filt):
    """Label rules fed by the engine's own stage 1 (logits in): labels come from a float argmax,
    so the comparison feeds the oracle the ENGINE's stage-1 outputs and demands exact maps."""
    import semantic_search_amd as ssa
    h, w, c, t = 480, 640, 40, 3.2
    lut = scenes.schema_lut(c)
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate=rule, seg_model_n_classes=c, filter_goal_prob=filt)
    ref = mo.OracleMapper(schema_lut=lut, path_overlay=False, **kw)
    eng = ssa.MapEngine(1, (h, w), max_map_hw=(480, 480), **kw)
    eng.set_schema(lut)
    for k, v in attrs.items():
        setattr(ref, k, v)
        setattr(eng.params, k, v)
    ref.reset(BOUNDS_480)
    maps = eng.reset_env(0, BOUNDS_480)
    rng, prng = np.random.default_rng(5), np.random.default_rng(6)
    for i in range(3):
        frame = scenes.uniform_frame(rng, h, w, c, t, lut=lut, with_probs=False)
        agent, cam = scenes.make_pose(7.2, 0.0, 7.2, prng.uniform(0, 2 * np.pi))
        depth = _to_dev(frame["depth"][None], np.float32)
        sem = torch.empty((1, h, w), dtype=torch.int32, device="cuda")
        unc = torch.empty((1, h, w), dtype=torch.float32, device="cuda")
        maxp = torch.empty((1, h, w), dtype=torch.float32, device="cuda")
        eng.step(depth, agent[None], cam[None], [3], logits=_to_dev(frame["logits"][None], np.float32),
                 temperature=t, semantic=sem, semantic_unc=unc, semantic_max_prob=maxp,
                 semantic_gt=_to_dev(frame["semantic_gt"][None], np.int32))
        found, status, n_upd = eng.results()
        obs = {"depth": frame["depth"].copy(), "semantic_gt": frame["semantic_gt"], "objectgoal": np.array([0]),
               "agent_position_and_rotation": agent, "camera_position_and_rotation": cam,
               "semantic": sem[0].cpu().numpy(), "semantic_unc": unc[0].cpu().numpy(),
               "semantic_max_prob": maxp[0].cpu().numpy()}
        found_ref = ref.update(obs)
        pix, _ = eng.update_list(0, int(n_upd[0]))
        assert np.array_equal(pix, ref.last_update_pixels)
        assert bool(found[0]) == bool(found_ref), f"frame {i}"
        assert np.array_equal(maps["semantic_map"].cpu().numpy().view(np.uint32), ref.semantic_map), f"frame {i}"
    for nm in ("height_map", "gt_map", "goal_map", "background_map", "hits", "views", "unc_map"):
        if nm in maps and getattr(ref, nm, None) is not None:
            got = maps[nm].cpu().numpy()
            if got.dtype == np.int32:
                got = got.view(np.uint32)
            assert np.array_equal(got, getattr(ref, nm)), nm
    eng.close()


@pytest.mark.parametrize("n_envs,hw", [(3, (96, 128)), (18, (96, 128)), (3, (45, 67))])
def test_batched_engine_equals_independent_engines(n_envs, hw):
    """E envs with different map sizes and poses in one call == E engines of one env each.
    (Up to 16 envs the poses travel as kernel arguments, above through a staging copy.  With the
    45x67 frames an env's pixels do not start at a multiple of four in the batch.)"""
    import semantic_search_amd as ssa
    (h, w), c = hw, 8
    lut = scenes.small_lut(c)
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate="height_map_avg_prob", seg_model_n_classes=c)
    base = [np.array([0, 0, 0, 4.5, 2.5, 4.5]), np.array([-1.0, 0, -2.0, 5.0, 2.5, 3.1]),
            np.array([0.3, 0, 0.2, 3.0, 2.5, 6.0])]
    bounds = [base[e % 3] for e in range(n_envs)]
    goals = [3 + (e % 2) for e in range(n_envs)]
    big = ssa.MapEngine(n_envs, (h, w), max_map_hw=(200, 200), **kw)
    big.set_schema(lut)
    singles = []
    for e in range(n_envs):
        big.reset_env(e, bounds[e])
        s = ssa.MapEngine(1, (h, w), max_map_hw=(200, 200), **kw)
        s.set_schema(lut)
        s.reset_env(0, bounds[e])
        singles.append(s)
    rng = np.random.default_rng(3)
    for i in range(3):
        frames, ag, cam = [], [], []
        for e in range(n_envs):
            a, cm = scenes.make_pose(1.5 + 0.2 * (e % 5), 0.0, 1.2 + 0.3 * i, rng.uniform(0, 6.28))
            frames.append(scenes.room_frame(rng, h, w, c, 1.0, cm, lut=lut))
            ag.append(a)
            cam.append(cm)
        depth = np.stack([f["depth"] for f in frames])
        probs = np.stack([f["semantic_prob"] for f in frames])
        gt = np.stack([f["semantic_gt"] for f in frames])
        for e in range(1, n_envs, 3):                          # every third env leaves its predicted cell box
            far = rng.random((h, w)) < 0.03                    # (depths beyond depth_max): scanned, the others binned
            depth[e][far] = rng.uniform(5.5, 11.0, int(far.sum())).astype(np.float32)
        depth[:, 0, :3] = 0.5                                  # sentinels, zeroed in place by the step
        d_big = _to_dev(depth, np.float32)
        big.step(d_big, np.stack(ag), np.stack(cam), goals,
                 probs=_to_dev(probs, np.float32), semantic_gt=_to_dev(gt, np.int32))
        fb, sb, nb = big.results()
        for e in range(n_envs):
            d_one = _to_dev(depth[e:e + 1], np.float32)
            singles[e].step(d_one, ag[e][None], cam[e][None], [goals[e]],
                            probs=_to_dev(probs[e:e + 1], np.float32), semantic_gt=_to_dev(gt[e:e + 1], np.int32))
            f1, s1_, n1 = singles[e].results()
            assert (fb[e], sb[e], nb[e]) == (f1[0], s1_[0], n1[0])
            assert torch.equal(d_big[e], d_one[0]) and (d_one[0, 0, :3] == 0).all()
    for e in range(n_envs):
        for nm, t in big.maps[e].items():
            assert torch.equal(t, singles[e].maps[0][nm]), (e, nm)
    big.close()
    for s in singles:
        s.close()


def test_benchmark_configuration_replicated_envs_agree():
    """BASELINE configs[1] at full size (height_map_log_odds, 16 envs, 640x480, 40 classes, logits
    in): envs fed the same frame and pose must end bit-identical, a one-env engine fed that frame
    must equal them, and envs fed another frame must differ."""
    import semantic_search_amd as ssa
    h, w, c, temp, E = 480, 640, 40, 3.2, 16
    lut = scenes.schema_lut(c)
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate="height_map_log_odds", seg_model_n_classes=c)
    big = ssa.MapEngine(E, (h, w), max_map_hw=ssa.map_size(BOUNDS_480, 0.03), **kw)
    one = ssa.MapEngine(1, (h, w), max_map_hw=ssa.map_size(BOUNDS_480, 0.03), **kw)
    for eng in (big, one):
        eng.set_schema(lut)
        eng.params.entropy_threshold = 0.0171
    for e in range(E):
        big.reset_env(e, BOUNDS_480)
    one.reset_env(0, BOUNDS_480)
    rng, prng = np.random.default_rng(61), np.random.default_rng(62)
    for i in range(2):
        fa = scenes.uniform_frame(rng, h, w, c, temp, lut=lut, with_probs=False)
        fb = scenes.uniform_frame(rng, h, w, c, temp, lut=lut, with_probs=False)
        pa = scenes.make_pose(7.2, 0.0, 7.2, prng.uniform(0, 2 * np.pi))
        pb = scenes.make_pose(7.0, 0.0, 7.4, prng.uniform(0, 2 * np.pi))
        pick = [fa if e < 8 else fb for e in range(E)]
        pose = [pa if e < 8 else pb for e in range(E)]
        depth = _to_dev(np.stack([f["depth"] for f in pick]), np.float32)
        logits = _to_dev(np.stack([f["logits"] for f in pick]), np.float32)
        gt = _to_dev(np.stack([f["semantic_gt"] for f in pick]), np.int32)
        sem = torch.empty((E, h, w), dtype=torch.int32, device="cuda")
        unc = torch.empty((E, h, w), dtype=torch.float32, device="cuda")
        maxp = torch.empty((E, h, w), dtype=torch.float32, device="cuda")
        big.step(depth, np.stack([p[0] for p in pose]), np.stack([p[1] for p in pose]), [3] * E, logits=logits,
                 temperature=temp, semantic=sem, semantic_unc=unc, semantic_max_prob=maxp, semantic_gt=gt)
        one.step(depth[:1].clone(), pa[0][None], pa[1][None], [3], logits=logits[:1].contiguous(), temperature=temp,
                 semantic=sem[:1].clone(), semantic_unc=unc[:1].clone(), semantic_max_prob=maxp[:1].clone(),
                 semantic_gt=gt[:1].contiguous())
        found, status, n_upd = big.results()
        assert not status.any()
        assert len(set(n_upd[:8].tolist())) == 1 and len(set(n_upd[8:].tolist())) == 1
        assert n_upd[0] == one.results()[2][0]
        assert torch.equal(sem[0], sem[7]) and torch.equal(unc[8], unc[15])
    for nm, t in big.maps[0].items():
        for e in range(1, 8):
            assert torch.equal(t, big.maps[e][nm]), (e, nm)
        assert torch.equal(t, one.maps[0][nm]), nm
        for e in range(9, 16):
            assert torch.equal(big.maps[8][nm], big.maps[e][nm]), (e, nm)
    assert not torch.equal(big.maps[0]["semantic_map"], big.maps[8]["semantic_map"])
    big.close()
    one.close()


def test_index_below_minus_dim_is_reported_like_numpy():
    """A selected cell with index < -dim makes the reference raise IndexError (mapper.py:338)."""
    import semantic_search_amd as ssa
    h, w = 48, 64
    m = ssa.SemanticMapper(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0))
    m.reset(np.array([4.0, 0.0, 4.0, 4.6, 2.5, 4.6]))      # 20 x 20 cells; camera 3 m outside
    agent, cam = scenes.make_pose(1.0, 0.0, 1.0, np.pi + 0.75)
    frame = scenes.room_frame(np.random.default_rng(0), h, w, 0, 1.0, cam)
    obs = scenes.make_obs(frame, agent, cam)
    ref = mo.OracleMapper(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), path_overlay=False)
    ref.reset(np.array([4.0, 0.0, 4.0, 4.6, 2.5, 4.6]))
    with pytest.raises(IndexError):
        ref.update(scenes.make_obs(frame, agent, cam))
    with pytest.raises(IndexError):
        m.update_map(obs, None)
    # the reference raises at its first indexed statement, before any map is written
    assert not m.semantic_map.any() and not m.semantic_map_color.any() and np.isneginf(m.height_map).all()


def test_index_below_minus_dim_under_a_taller_rejected_pixel_is_no_error():
    """numpy raises only for a SELECTED cell (mapper.py:331-338): a pixel below -dim that passes the
    height test does not raise when a pixel above the cut-off is the tallest of its cell.  Flat depth
    bands seen level from outside the map: every image column of a band is one cell, and the band
    that straddles the horizon has its upper rows above the cut-off.  The engine agrees with the
    oracle on the report, on the update set and on the maps."""
    import semantic_search_amd as ssa
    h, w = 48, 64
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0))
    cases = []
    for trial in range(16):     # a 20 x 20 map seen from up to 4 m outside: candidates, some of them real errors
        rng = np.random.default_rng(trial)
        agent, cam = scenes.make_pose(rng.uniform(0, 4.4), 0.0, rng.uniform(0, 4.4), rng.uniform(0, 2 * np.pi), pitch=0.0)
        depth = np.full((h, w), np.float32(rng.uniform(0.7, 4.5)))
        if trial % 2:
            depth[2 * h // 3:] = np.float32(rng.uniform(1.5, 4.5))
        cases.append((np.array([4.0, 0.0, 4.0, 4.6, 2.5, 4.6]), (20, 20), agent, cam, depth, rng))
    for k in range(7, 14):      # a 100 x 100 map: the near band below -dim under its own upper rows, the far band inside
        rng = np.random.default_rng(100 + k)
        agent, cam = scenes.make_pose(-3.0, 0.0, 3.5, k * np.pi / 8, pitch=0.0)
        depth = np.full((h, w), np.float32(1.0))
        depth[2 * h // 3:] = np.float32(4.0)
        if k % 2:               # invalid depths sit in the camera's cell, itself below -dim here
            depth[rng.random((h, w)) < 0.05] = np.float32(0.0)
        cases.append((np.array([2.0, 0.0, 2.0, 5.0, 2.5, 5.0]), (100, 100), agent, cam, depth, rng))
    suppressed = both = raised_frames = 0
    for bounds, (mh, mw), agent, cam, depth, rng in cases:
        frame = scenes.uniform_frame(rng, h, w, 0, 1.0)
        frame["depth"] = depth
        rows, cols, heights, valid = mo.project_frame(depth.copy(), 59.0, (0.5, 5.0), cam, bounds[:3], 0.03)
        cand = valid & (heights < cam[1] + 0.1) & (rows < mh) & (cols < mw) & ((rows < -mh) | (cols < -mw))
        ref = mo.OracleMapper(path_overlay=False, **kw)
        ref.reset(bounds)
        eng = ssa.MapEngine(1, (h, w), max_map_hw=(mh, mw), **kw)
        maps = eng.reset_env(0, bounds)
        raised = False
        try:
            ref.update(scenes.make_obs(frame, agent, cam))
        except IndexError:
            raised = True
        eng.step(torch.from_numpy(depth[None].copy()).cuda(), agent[None], cam[None], [3],
                 semantic=torch.from_numpy(frame["semantic"][None].astype(np.int32)).cuda())
        found, status, n_upd = eng.results()
        assert bool(status[0] & 1) == raised
        if raised:
            raised_frames += 1
            assert n_upd[0] == 0 and np.isneginf(maps["height_map"].cpu().numpy()).all()
        else:
            suppressed += int(cand.any())
            both += int(cand.any() and n_upd[0] > 0)
            pix, _ = eng.update_list(0, int(n_upd[0]))
            assert np.array_equal(np.sort(pix), np.sort(ref.last_update_pixels))
            assert np.array_equal(maps["height_map"].cpu().numpy(), ref.height_map)
            assert np.array_equal(maps["semantic_map"].cpu().numpy().view(np.uint32), ref.semantic_map)
        eng.close()
    assert suppressed >= 8 and both >= 3 and raised_frames >= 3, (suppressed, both, raised_frames)


def test_frame_without_any_valid_depth_updates_nothing():
    import semantic_search_amd as ssa
    m = ssa.SemanticMapper(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0))
    m.reset(np.array([0.0, 0.0, 0.0, 3.0, 2.5, 3.0]))
    agent, cam = scenes.make_pose(1.5, 0.0, 1.5, 0.3)
    obs = {"depth": np.full((48, 64), 5.0, np.float32), "semantic": np.ones((48, 64), np.int32),
           "agent_position_and_rotation": agent, "camera_position_and_rotation": cam,
           "objectgoal": np.array([1])}
    _, found = m.update_map(obs, None)
    assert found is None and m.last_n_updates == 0
    assert not obs["depth"].any() and not m.semantic_map.any()
    assert np.isneginf(m.height_map).all()


def test_update_map_takes_torch_tensors_like_numpy_arrays():
    """Observations that are torch tensors (host or GPU) give the maps the numpy arrays give; the
    sentinels of a GPU depth tensor are zeroed in place like those of a numpy depth image."""
    import torch
    import semantic_search_amd as ssa
    lut = scenes.small_lut(8)
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate="height_map_log_odds", seg_model_n_classes=8)
    bounds = np.array([0.0, -0.2, 0.0, 4.5, 2.5, 4.5])
    mappers = []
    for _ in range(2):
        m = ssa.SemanticMapper(**kw)
        m.map_seg_to_schema = lambda ids: lut[ids]
        m.reset(bounds)
        mappers.append(m)
    rng = np.random.default_rng(5)
    for i in range(3):
        agent, cam = scenes.make_pose(1.5 + 0.2 * i, 0.0, 1.2, 0.8 - 0.3 * i)
        frame = scenes.room_frame(rng, 96, 128, 8, 1.0, cam, lut=lut)
        frame["depth"][0, :5] = 0.5                      # sentinels (mapper.py:282-283)
        obs_np = scenes.make_obs(frame, agent, cam, goal_index=0)
        obs_t = dict(obs_np)
        for k in ("depth", "semantic_prob", "semantic_gt"):
            obs_t[k] = torch.from_numpy(obs_np[k].copy())
            if k != "semantic_gt":
                obs_t[k] = obs_t[k].cuda()                # one observation stays on the host
        ego_a, found_a = mappers[0].update_map(obs_np)
        ego_b, found_b = mappers[1].update_map(obs_t)
        assert found_a == found_b
        assert np.array_equal(obs_np["depth"], obs_t["depth"].cpu().numpy())
        assert (obs_np["depth"][0, :5] == 0).all()
        for a, b in zip(ego_a, ego_b):
            assert np.array_equal(a, b)
    for name in ("semantic_map", "height_map", "gt_map", "log_odds_seg_map", "unc_map", "semantic_map_color"):
        assert np.array_equal(getattr(mappers[0], name), getattr(mappers[1], name), equal_nan=True), name


def test_host_arrays_staged_whole_in_pieces_or_pageable_give_the_same_maps():
    """A full-size frame (49 MB of class probabilities) fed as numpy arrays: the page-locked staging
    buffer filled in one piece, in overlapped pieces (the default) and plain pageable copies end in
    the same maps, over two frames (the second reuses the staging buffers)."""
    import semantic_search_amd as ssa
    c = 40
    lut = scenes.schema_lut(c)
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate="height_map_avg_prob", seg_model_n_classes=c)
    bounds = np.array([0.0, -0.2, 0.0, 4.5, 2.5, 4.5])
    mappers = []
    for pinned, chunks in ((True, 1), (True, 4), (True, 7), (False, 1)):
        m = ssa.SemanticMapper(**kw)
        m.map_seg_to_schema = lambda ids: lut[ids]
        m.pinned_staging, m.staging_chunks = pinned, chunks
        m.reset(bounds)
        mappers.append(m)
    rng = np.random.default_rng(17)
    for i in range(2):
        agent, cam = scenes.make_pose(1.5 + 0.3 * i, 0.0, 1.2, 0.8 - 0.3 * i)
        frame = scenes.room_frame(rng, 480, 640, c, 1.0, cam, lut=lut)
        outs = [m.update_map(scenes.make_obs(frame, agent, cam, goal_index=0)) for m in mappers]
        for ego, found in outs[1:]:
            assert found == outs[0][1]
            for a, b in zip(ego, outs[0][0]):
                assert np.array_equal(a, b)
    assert mappers[0].semantic_map.any()
    for m in mappers[1:]:
        for name in ("semantic_map", "height_map", "gt_map", "avg_prob_seg_map", "avg_prob_weight_map", "unc_map"):
            assert np.array_equal(getattr(m, name), getattr(mappers[0], name), equal_nan=True), name


@pytest.mark.parametrize("rule", ["height_map_log_odds", "height_map_prob_hv", "height_map_goal_decay"])
def test_mapper_episodes_with_different_map_bounds(rule):
    """One SemanticMapper over three episodes, like an env that resets it per scene (env.py:377):
    a small map, a larger one (the engine is rebuilt for it), then a small one again.  After every
    frame of every episode the maps equal those of an oracle that was reset the same way: nothing
    of an earlier episode (overlay lists, trail, per-step scratch) leaks into the next."""
    import semantic_search_amd as ssa
    lut = scenes.small_lut(8)
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate=rule, seg_model_n_classes=8)
    m = ssa.SemanticMapper(**kw)
    m.map_seg_to_schema = lambda ids: lut[ids]
    ref = mo.OracleMapper(schema_lut=lut, **kw)
    for o in (m, ref):
        if "hv" in rule:
            o.hv_min_views, o.hv_ratio = 1, 0.4
    rng = np.random.default_rng(21)
    episodes = [np.array([0.0, -0.2, 0.0, 4.5, 2.5, 4.5]), np.array([-1.5, -0.2, -2.0, 6.5, 2.5, 7.0]),
                np.array([0.4, -0.2, 0.3, 4.0, 2.5, 4.2])]
    for ep, bounds in enumerate(episodes):
        m.reset(bounds)
        ref.reset(bounds)
        for i in range(4):
            agent, cam = scenes.make_pose(1.0 + 0.2 * i, 0.0, 1.0 + 0.2 * i, np.pi + 0.75 + 0.05 * ep)
            frame = scenes.room_frame(rng, 96, 128, 8, 1.0, cam, lut=lut, box=(2.0, 2.0, 2.9, 2.8, 0.5),
                                      speckle=0.01, sharp=12.0)
            obs = scenes.make_obs(frame, agent, cam, goal_index=0)
            want_found = ref.update({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in obs.items()})
            _, found = m.update_map(obs)
            assert bool(found) == bool(want_found), (ep, i, "found")
            assert m.semantic_map.shape == ref.semantic_map.shape
            assert np.array_equal(m.semantic_map, ref.semantic_map), (ep, i, "semantic_map")
            assert np.array_equal(m.semantic_map_color, ref.semantic_map_color), (ep, i, "colour")
            assert np.array_equal(m.height_map, ref.height_map), (ep, i, "height_map")
            if "hv" in rule:
                assert np.array_equal(m.hits, ref.hits) and np.array_equal(m.views, ref.views), (ep, i)
            if "goal_decay" in rule:
                assert np.array_equal(m.goal_map, ref.goal_map), (ep, i, "goal_map")
            if "log_odds" in rule:
                parity.close_rows(m.log_odds_seg_map, ref.log_odds_seg_map, "log_odds_seg_map")


@pytest.mark.parametrize("rule", ["height_map_prob_hv", "height_map_avg_prob", "height_map_goal_decay"])
def test_one_env_of_a_batch_is_reset_while_the_others_go_on(rule):
    """Vectorised envs end their episodes at different times: env 1 of three is reset (new bounds)
    after two steps while envs 0 and 2 keep their maps.  Every env equals an engine of its own that
    saw the same frames and the same reset."""
    import semantic_search_amd as ssa
    h, w, c, n_envs = 96, 128, 8, 3
    lut = scenes.small_lut(c)
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate=rule, seg_model_n_classes=c)
    bounds = [np.array([0, 0, 0, 4.5, 2.5, 4.5]), np.array([-1.0, 0, -2.0, 5.0, 2.5, 3.1]),
              np.array([0.3, 0, 0.2, 3.0, 2.5, 6.0])]
    new_bounds = np.array([0.5, 0, 0.5, 5.5, 2.5, 5.0])
    big = ssa.MapEngine(n_envs, (h, w), max_map_hw=(220, 220), **kw)
    big.set_schema(lut)
    singles = []
    for e in range(n_envs):
        big.reset_env(e, bounds[e])
        s = ssa.MapEngine(1, (h, w), max_map_hw=(220, 220), **kw)
        s.set_schema(lut)
        s.reset_env(0, bounds[e])
        singles.append(s)
    for o in [big.params] + [s.params for s in singles]:
        o.hv_min_views, o.hv_ratio = 1, 0.4
    rng = np.random.default_rng(4)
    prob_rule = "avg_prob" in rule
    for i in range(5):
        if i == 2:
            big.reset_env(1, new_bounds)
            singles[1].reset_env(0, new_bounds)
        frames, ag, cam = [], [], []
        for e in range(n_envs):
            a, cm = scenes.make_pose(1.0 + 0.2 * i, 0.0, 1.0 + 0.15 * i + 0.1 * e, np.pi + 0.75)
            frames.append(scenes.room_frame(rng, h, w, c, 1.0, cm, lut=lut, box=(2.0, 2.0, 2.9, 2.8, 0.5),
                                            speckle=0.01, sharp=12.0))
            ag.append(a)
            cam.append(cm)
        stack = lambda key, dt: np.stack([f[key] for f in frames]).astype(dt)
        def feed(eng, sl, goals):
            kwargs = dict(semantic_gt=_to_dev(stack("semantic_gt", np.int32)[sl], np.int32))
            if prob_rule:
                kwargs["probs"] = _to_dev(stack("semantic_prob", np.float32)[sl], np.float32)
            else:
                for key, dt in (("semantic", np.int32), ("semantic_unc", np.float32), ("semantic_max_prob", np.float32)):
                    kwargs[key] = _to_dev(stack(key, dt)[sl], dt)
            eng.step(_to_dev(stack("depth", np.float32)[sl], np.float32), np.stack(ag)[sl], np.stack(cam)[sl], goals, **kwargs)
            return eng.results()
        fb, sb, nb = feed(big, slice(0, n_envs), [3] * n_envs)
        for e in range(n_envs):
            f1, s1_, n1 = feed(singles[e], slice(e, e + 1), [3])
            assert (fb[e], sb[e], nb[e]) == (f1[0], s1_[0], n1[0]), (i, e)
            for nm, t in big.maps[e].items():
                assert torch.equal(t, singles[e].maps[0][nm]), (i, e, nm)
    big.close()
    for s in singles:
        s.close()


@pytest.mark.parametrize("rule,c", [("height_map_log_odds", 40), ("height_map_w2_avg_prob", 40), ("height_map_avg_prob", 8),
                                    ("height_map_w2_avg_prob", 150), ("height_map_log_odds", 150),
                                    ("height_map_log_odds", 37), ("height_map_avg_prob", 33), ("height_map_w_avg_prob", 53),
                                    ("height_map_w2_avg_prob", 81), ("height_map_avg_prob", 12), ("height_map_log_odds", 6)])
def test_direct_update_equals_rows_and_update_kernel(rule, c, monkeypatch):
    """The two homes of the probability rules' update for the class counts with a register kernel: inside the streaming kernel
    (default) and as staged rows + k_update_prob (SSMAP_DIRECT=0 when the engine is created).  Same
    update_row; integer maps and the log-odds sums are bit-identical, the averages and their entropy
    weights agree to 1e-6 (the entropy sums run over another class-to-lane layout)."""
    import semantic_search_amd as ssa
    lut = scenes.schema_lut(c) if c >= 33 else scenes.small_lut(c)
    bounds = np.array([0.0, -0.2, 0.0, 4.5, 2.5, 4.5])
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate=rule, seg_model_n_classes=c, max_map_hw=ssa.map_size(bounds, 0.03))
    engs = []
    # the default, staged rows written by the register kernel (exact class counts only; the others
    # fall to the generic kernel), staged rows written by the generic LDS-column kernel
    for direct, generic in (("1", "0"), ("0", "0"), ("1", "1")):
        monkeypatch.setenv("SSMAP_DIRECT", direct)
        monkeypatch.setenv("SSMAP_GENERIC", generic)
        eng = ssa.MapEngine(2, (96, 128), **kw)
        eng.set_schema(lut)
        for e in range(2):
            eng.reset_env(e, bounds)
        engs.append(eng)
    rng = np.random.default_rng(11)
    for t in range(4):
        frames, ag, cam = [], [], []
        for e in range(2):
            a, cm = scenes.make_pose(1.2 + 0.3 * e, 0.0, 1.0 + 0.3 * t, np.pi + 0.75 + 0.1 * e)
            frames.append(scenes.room_frame(rng, 96, 128, c, 1.0, cm, lut=lut, box=(2.0, 2.0, 2.9, 2.8, 0.5), sharp=4.0))
            ag.append(a)
            cam.append(cm)
        dev = lambda key, dt: torch.from_numpy(np.stack([f[key] for f in frames]).astype(dt)).cuda()
        res = []
        for eng in engs:
            eng.step(dev("depth", np.float32), np.stack(ag), np.stack(cam), [3, 3],
                     probs=dev("semantic_prob", np.float32), semantic_gt=dev("semantic_gt", np.int32))
            res.append(eng.results())
        for other_res in res[1:]:
            for x, y in zip(res[0], other_res):
                assert np.array_equal(x, y), t
        assert res[0][2].min() > 100
    for e, other_eng in ((e, o) for e in range(2) for o in engs[1:]):
        for nm, m in engs[0].maps[e].items():
            other = other_eng.maps[e][nm]
            if m.dtype.is_floating_point and "log_odds" not in rule and nm in ("avg_prob_seg_map", "avg_prob_weight_map", "unc_map"):
                # (w_avg_prob: the weight is 1 - entropy, a difference of nearly equal numbers whose entropy
                # sums run over two class-to-lane layouts: the parity bar itself, 1e-5 + 1e-6, applies)
                tol = dict(rtol=1e-5, atol=1e-6) if "w_avg" in rule else dict(rtol=1e-6, atol=1e-7)
                assert torch.allclose(m, other, equal_nan=True, **tol), nm
            else:
                assert torch.equal(m, other), nm
    for eng in engs:
        eng.close()
