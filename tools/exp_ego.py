"""Time of the ego-map kernel for 16 envs (run on the GPU box)."""
import os, sys, math
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semantic_search_amd as ssa

def main():
    E, span = 16, 14.39
    eng = ssa.MapEngine(E, (480, 640), num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0),
                        probablistic_mapping=True, prob_aggregate="height_map_log_odds", seg_model_n_classes=40,
                        max_map_hw=ssa.map_size([0, 0, 0, span, 3.0, span], 0.03))
    for e in range(E):
        maps = eng.reset_env(e, [0, 0, 0, span, 3.0, span])
        maps["semantic_map"].random_(0, 9)
    rng = np.random.default_rng(0)
    poses = np.zeros((E, 7)); poses[:, 0] = poses[:, 2] = span / 2
    yaw = rng.uniform(-math.pi, math.pi, E)
    poses[:, 3] = np.cos(yaw / 2); poses[:, 5] = np.sin(yaw / 2)
    goals = np.full(E, 3, dtype=np.int32)
    aux = np.eye(12, dtype=np.float32)[rng.integers(0, 12, E)]
    for _ in range(3):
        eng.ego_maps(poses, goals, aux)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        eng.ego_maps(poses, goals, aux)
    b.record(); torch.cuda.synchronize()
    print(f"ego maps, {E} envs: {a.elapsed_time(b) / 50 * 1e3:.1f} us per call (includes four output allocations)")
main()
