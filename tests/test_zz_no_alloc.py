"""ssmap_step_noalloc (per-step scratch in the library's file-scope device array) against
ssmap_step (scratch allocated by ssmap_create), through the C ABI: same inputs, bit-identical maps,
update lists and results, on two calls in a row of which the second is smaller (the array still
holds the first call's contents)."""
import numpy as np
import pytest
import torch

from tests import scenes

pytestmark = pytest.mark.gpu


def _dev(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


def _pair(rule, n_envs, hw, c, bounds, filt=None):
    import semantic_search_amd as ssa
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate=rule, seg_model_n_classes=c, filter_goal_prob=filt)
    engines = []
    for _ in range(2):
        eng = ssa.MapEngine(n_envs, hw, max_map_hw=ssa.map_size(bounds, 0.03), **kw)
        eng.set_schema(scenes.small_lut(c))
        for e in range(n_envs):
            eng.reset_env(e, bounds)
        engines.append(eng)
    return engines


def _drive(engines, rule, n_envs, hw, c, n_frames, seed):
    h, w = hw
    lut = scenes.small_lut(c)
    rng = np.random.default_rng(seed)
    prob_rule = "avg_prob" in rule or "log_odds" in rule
    for i in range(n_frames):
        frames, ag, cam = [], [], []
        for e in range(n_envs):
            a, cm = scenes.make_pose(1.2 + 0.2 * e, 0.0, 1.0 + 0.25 * i, np.pi + 0.75 + 0.1 * e)
            frames.append(scenes.room_frame(rng, h, w, c, 1.0, cm, lut=lut, box=(2.0, 2.0, 2.9, 2.8, 0.5),
                                            speckle=0.01, sharp=12.0))
            ag.append(a)
            cam.append(cm)
        stack = lambda key, dt: np.stack([f[key] for f in frames]).astype(dt)
        for k, eng in enumerate(engines):
            kwargs = dict(semantic_gt=_dev(stack("semantic_gt", np.int32), np.int32))
            if prob_rule:
                kwargs["probs"] = _dev(stack("semantic_prob", np.float32), np.float32)
            else:
                for key, dt in (("semantic", np.int32), ("semantic_unc", np.float32), ("semantic_max_prob", np.float32)):
                    kwargs[key] = _dev(stack(key, dt), dt)
            eng.step(_dev(stack("depth", np.float32), np.float32), np.stack(ag), np.stack(cam), [3] * n_envs,
                     static_scratch=(k == 1), **kwargs)
        ra, rb = engines[0].results(), engines[1].results()
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y), f"frame {i}: results"
        assert ra[2].min() > 50, "the frames must update some cells"
        for e in range(n_envs):
            la = engines[0].update_list(e, int(ra[2][e]))
            lb = engines[1].update_list(e, int(rb[2][e]))
            assert np.array_equal(la[0], lb[0]) and np.array_equal(la[1], lb[1]), (i, e, "update list")
    for e in range(n_envs):
        for nm, t in engines[0].maps[e].items():
            assert torch.equal(t, engines[1].maps[e][nm]), (e, nm)


def test_noalloc_step_equals_step_on_two_calls_the_second_smaller():
    bounds = np.array([0.0, -0.2, 0.0, 4.5, 2.5, 4.5])
    first = _pair("height_map_avg_prob", 3, (96, 128), 8, bounds)
    _drive(first, "height_map_avg_prob", 3, (96, 128), 8, 3, seed=1)
    # a smaller problem right after, another rule (goal_decay keeps local maps in the scratch): the
    # array still holds the records, lists and rows of the calls above
    second = _pair("height_map_goal_decay", 1, (48, 64), 8, bounds, filt=0.35)
    _drive(second, "height_map_goal_decay", 1, (48, 64), 8, 4, seed=2)
    # and the first engines again after the smaller calls
    _drive(first, "height_map_avg_prob", 3, (96, 128), 8, 2, seed=3)
    for eng in first + second:
        eng.close()


def test_noalloc_step_refuses_an_engine_whose_scratch_exceeds_the_array():
    import semantic_search_amd as ssa
    from semantic_search_amd import _capi
    h, w, c, E = 480, 640, 40, 2
    bounds = np.array([0.0, 0.0, 0.0, 14.39, 3.0, 14.39])
    eng = ssa.MapEngine(E, (h, w), num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0),
                        probablistic_mapping=True, prob_aggregate="height_map_log_odds", seg_model_n_classes=c,
                        max_map_hw=ssa.map_size(bounds, 0.03))
    eng.set_schema(scenes.schema_lut(c))
    maps = [eng.reset_env(e, bounds) for e in range(E)]
    before = [m["log_odds_seg_map"].clone() for m in maps]
    ag, cam = scenes.make_pose(7.2, 0.0, 7.2, 0.3)
    depth = torch.full((E, h, w), 2.0, dtype=torch.float32, device="cuda")
    with pytest.raises(_capi.EngineError, match="exceeds the static array"):
        eng.step(depth, np.stack([ag] * E), np.stack([cam] * E), [3] * E,
                 probs=torch.full((E, c, h, w), 1.0 / c, dtype=torch.float32, device="cuda"),
                 semantic_gt=torch.zeros((E, h, w), dtype=torch.int32, device="cuda"), static_scratch=True)
    torch.cuda.synchronize()
    for m, b in zip(maps, before):
        assert torch.equal(m["log_odds_seg_map"], b), "a refused call must not touch the maps"
    eng.close()
