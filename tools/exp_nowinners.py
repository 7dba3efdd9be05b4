"""Streaming kernel inside a full step when no pixel owns a map cell (all depths invalid): what the
fused variant costs without any probability row to hand over.  Read k_stream_reg<40, 2> from a
rocprofv3 --kernel-trace --stats run of this script."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semantic_search_amd as ssa
import bench
from tests import scenes

def main():
    dev = torch.device("cuda:0")
    E, C, span = 16, 40, 14.39
    eng = ssa.MapEngine(E, (bench.H, bench.W), num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0),
                        probablistic_mapping=True, prob_aggregate="height_map_log_odds", seg_model_n_classes=C,
                        max_map_hw=ssa.map_size([0, 0, 0, span, 3.0, span], 0.03), device=dev)
    eng.params.entropy_threshold = 0.0171
    eng.set_schema(scenes.schema_lut(C))
    for e in range(E):
        eng.reset_env(e, [0, 0, 0, span, 3.0, span])
    depth, logits, gt = bench.synth_frames(E, C, 0, dev)
    depth.zero_()
    sem = torch.empty((E, bench.H, bench.W), dtype=torch.int32, device=dev)
    unc = torch.empty((E, bench.H, bench.W), dtype=torch.float32, device=dev)
    maxp = torch.empty((E, bench.H, bench.W), dtype=torch.float32, device=dev)
    goals = np.full(E, 3, dtype=np.int32)
    prng = np.random.default_rng(7)
    for i in range(30):
        ag, cam = bench.make_poses(E, span / 2, prng)
        eng.step(depth, ag, cam, goals, logits=logits, temperature=3.2, semantic=sem,
                 semantic_unc=unc, semantic_max_prob=maxp, semantic_gt=gt)
    torch.cuda.synchronize()
    print("updates per env:", eng.results()[2][:4])
main()
