"""The benchmark configurations at their real per-GPU batch sizes (BASELINE.json configs[2..4]):

    height_map_w2_avg_prob   64 envs, 150 classes, T = 2.5, 480 x 480 maps
    height_map_goal_decay    64 envs (a shard of the 256), filter_goal_prob, 40 classes
    height_map_prob_hv      128 envs (a shard of the 1024), 960 x 960 maps, 40 classes

The oracle needs about a tenth of a second per frame and env, so at these sizes the check is a
property the path must have whatever the batch size: envs fed the same frames and poses end
bit-identical, equal to a one-env engine fed those frames (whose parity against the oracle the
other GPU tests pin), and different from the envs fed other frames.  Half of the envs get frame A,
the other half frame B, for two steps with new poses."""
import numpy as np
import pytest
import torch

from tests import scenes

pytestmark = pytest.mark.gpu

H, W = 480, 640


@pytest.mark.parametrize("rule,c,temp,E,span,attrs,filt", [
    ("height_map_w2_avg_prob", 150, 2.5, 64, 14.39, dict(entropy_threshold=0.4820), None),
    ("height_map_goal_decay", 40, 3.2, 64, 14.39, dict(goal_decay=0.8931, goal_mark=1.5408), 0.38158),
    ("height_map_prob_hv", 40, 3.2, 128, 28.79, dict(hv_ratio=0.887, hv_min_views=5, hv_limit_obj_view_dist=4.13219), None),
])
def test_full_batch_replicated_envs(rule, c, temp, E, span, attrs, filt):
    import semantic_search_amd as ssa
    bounds = np.array([0.0, 0.0, 0.0, span, 3.0, span])
    lut = scenes.schema_lut(c)
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate=rule, seg_model_n_classes=c, filter_goal_prob=filt)
    hw_map = ssa.map_size(bounds, 0.03)
    big = ssa.MapEngine(E, (H, W), max_map_hw=hw_map, **kw)
    ones = [ssa.MapEngine(1, (H, W), max_map_hw=hw_map, **kw) for _ in range(2)]
    for eng in [big] + ones:
        eng.set_schema(lut)
        for k, v in attrs.items():
            setattr(eng.params, k, v)
    for e in range(E):
        big.reset_env(e, bounds)
    for o in ones:
        o.reset_env(0, bounds)
    half = E // 2
    which = torch.tensor([0 if e < half else 1 for e in range(E)], device="cuda")
    rng, prng = np.random.default_rng(91), np.random.default_rng(92)
    mid = span / 2
    sem = torch.empty((E, H, W), dtype=torch.int32, device="cuda")
    unc = torch.empty((E, H, W), dtype=torch.float32, device="cuda")
    maxp = torch.empty((E, H, W), dtype=torch.float32, device="cuda")
    for step in range(2):
        frames = [scenes.uniform_frame(rng, H, W, c, temp, lut=lut, with_probs=False) for _ in range(2)]
        poses = [scenes.make_pose(mid, 0.0, mid, prng.uniform(0, 2 * np.pi)),
                 scenes.make_pose(mid - 0.2, 0.0, mid + 0.2, prng.uniform(0, 2 * np.pi))]
        pair = {k: torch.from_numpy(np.stack([f[k] for f in frames])).cuda()
                for k in ("depth", "logits", "semantic_gt")}
        depth = pair["depth"][which].contiguous()
        logits = pair["logits"][which].contiguous()
        gt = pair["semantic_gt"].to(torch.int32)[which].contiguous()
        ag = np.stack([poses[0 if e < half else 1][0] for e in range(E)])
        cam = np.stack([poses[0 if e < half else 1][1] for e in range(E)])
        big.step(depth, ag, cam, [3] * E, logits=logits, temperature=temp, semantic=sem, semantic_unc=unc,
                 semantic_max_prob=maxp, semantic_gt=gt)
        found, status, n_upd = big.results()
        assert not status.any()
        for k, o in enumerate(ones):
            s1 = torch.empty((1, H, W), dtype=torch.int32, device="cuda")
            u1 = torch.empty((1, H, W), dtype=torch.float32, device="cuda")
            m1 = torch.empty((1, H, W), dtype=torch.float32, device="cuda")
            o.step(pair["depth"][k:k + 1].clone(), poses[k][0][None], poses[k][1][None], [3],
                   logits=pair["logits"][k:k + 1].contiguous(), temperature=temp, semantic=s1, semantic_unc=u1,
                   semantic_max_prob=m1, semantic_gt=pair["semantic_gt"][k:k + 1].to(torch.int32).contiguous())
            f1, st1, n1 = o.results()
            lo, hi = (0, half) if k == 0 else (half, E)
            assert (n_upd[lo:hi] == n1[0]).all() and (found[lo:hi] == f1[0]).all(), (step, k)
            assert torch.equal(sem[lo], s1[0]) and torch.equal(sem[hi - 1], s1[0])
            assert torch.equal(unc[hi - 1], u1[0]) and torch.equal(maxp[lo], m1[0])
        del depth, logits, gt, pair
    for nm, t in big.maps[0].items():
        assert torch.equal(t, ones[0].maps[0][nm]), nm
        assert torch.equal(big.maps[half][nm], ones[1].maps[0][nm]), nm
        for e in range(1, half):
            assert torch.equal(t, big.maps[e][nm]), (e, nm)
        for e in range(half + 1, E):
            assert torch.equal(big.maps[half][nm], big.maps[e][nm]), (e, nm)
    assert not torch.equal(big.maps[0]["semantic_map"], big.maps[half]["semantic_map"])
    big.close()
    for o in ones:
        o.close()
