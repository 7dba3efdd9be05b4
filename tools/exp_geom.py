"""Exactness stress of the geometry kernel's verified division (GPU box): many random frames and
poses, cell indices and heights compared bit for bit with the C oracle."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semantic_search_amd as ssa
from oracle import mapper_oracle as mo
from tests import scenes
H, W = 480, 640
def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(123)
    m = ssa.SemanticMapper(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0))
    bounds = np.array([0.0, 0.0, 0.0, 14.39, 3.0, 14.39])
    ref = mo.OracleMapper(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0))
    bad = 0
    for i in range(n_frames):
        m.reset(bounds); ref.reset(bounds)
        depth = rng.uniform(0.5, 5.0, (H, W)).astype(np.float32)
        if i % 3 == 0:   # coarse quanta: many exact ties and exactly representable quotients
            depth = (np.round(depth * 8) / 8).astype(np.float32)
        sem = rng.integers(0, 9, (H, W)).astype(np.int32)
        a, c = scenes.make_pose(rng.uniform(3, 11), 0.0, rng.uniform(3, 11), rng.uniform(0, 6.3),
                                pitch=rng.uniform(-0.6, 0.1))
        obs = {"depth": depth.copy(), "semantic": sem, "agent_position_and_rotation": a,
               "camera_position_and_rotation": c, "objectgoal": np.array([0])}
        obs2 = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in obs.items()}
        m.update_map(obs, None); ref.update(obs2)
        pix, cells = m._engine.update_list(0, m.last_n_updates)
        ok = np.array_equal(pix, ref.last_update_pixels) and np.array_equal(m.height_map, ref.height_map) \
            and np.array_equal(m.semantic_map, ref.semantic_map)
        bad += not ok
    print(f"{n_frames} frames, mismatching frames: {bad}")
    assert bad == 0
main()
