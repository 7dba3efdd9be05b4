"""How far inside the 1e-5 bar the fused float maps are: worst error / tolerance per map, engine
against the oracle, over a few frames (run on the GPU box)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semantic_search_amd as ssa
from oracle import mapper_oracle as mo
from tests import scenes

RTOL, ATOL = 1e-5, 1e-6


def ratio(got, want, rows=False):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    fin = np.isfinite(want) & np.isfinite(got)
    scale = np.abs(want)
    if rows:
        scale = np.broadcast_to(np.max(np.where(np.isfinite(want), np.abs(want), 0.0), axis=-1, keepdims=True), want.shape)
    err = np.abs(np.where(fin, got - want, 0.0))
    return float((err / (ATOL + RTOL * scale))[fin].max())


for rule, c, hw in (("height_map_avg_prob", 40, (240, 320)), ("height_map_w2_avg_prob", 40, (240, 320)),
                    ("height_map_w_avg_prob", 40, (240, 320)), ("height_map_log_odds", 40, (240, 320)),
                    ("height_map_w2_avg_prob", 150, (120, 160)), ("height_map_log_odds", 150, (120, 160))):
    h, w = hw
    lut = scenes.schema_lut(c)
    bounds = np.array([0.0, -0.2, 0.0, 4.5, 2.5, 4.5])
    kw = dict(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
              prob_aggregate=rule, seg_model_n_classes=c)
    ref = mo.OracleMapper(schema_lut=lut, path_overlay=False, **kw)
    ref.reset(bounds)
    eng = ssa.MapEngine(1, (h, w), max_map_hw=ssa.map_size(bounds, 0.03), **kw)
    eng.set_schema(lut)
    maps = eng.reset_env(0, bounds)
    rng = np.random.default_rng(7)
    for i in range(6):
        agent, cam = scenes.make_pose(1.2 + 0.2 * i, 0.0, 1.0 + 0.1 * i, np.pi + 0.6 + 0.1 * i)
        frame = scenes.room_frame(rng, h, w, c, 1.0, cam, lut=lut, sharp=3.0)
        ref.update(scenes.make_obs(frame, agent, cam, goal_index=0))
        dev = lambda k, dt: torch.from_numpy(frame[k][None].astype(dt)).cuda()
        eng.step(dev("depth", np.float32), agent[None], cam[None], [3], probs=dev("semantic_prob", np.float32),
                 semantic_gt=dev("semantic_gt", np.int32))
    out = {}
    if "log_odds" in rule:
        out["log_odds_seg_map"] = ratio(maps["log_odds_seg_map"].cpu().numpy(), ref.log_odds_seg_map, rows=True)
    else:
        out["avg_prob_seg_map"] = ratio(maps["avg_prob_seg_map"].cpu().numpy(), ref.avg_prob_seg_map)
        out["avg_prob_weight_map"] = ratio(maps["avg_prob_weight_map"].cpu().numpy(), ref.avg_prob_weight_map)
        out["unc_map"] = ratio(maps["unc_map"].cpu().numpy(), ref.unc_map)
    print(f"{rule:26s} C={c:3d}: worst error / tolerance " + ", ".join(f"{k} {v:.3f}" for k, v in out.items()))
    eng.close()
