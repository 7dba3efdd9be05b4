"""Frames/s of the one-env SemanticMapper shim when it is fed host (numpy) arrays, i.e. with the
host-to-device copies of one frame's observations and the device-to-host copies of the four ego
maps inside the timed region.  This is the rate a caller of the reference's own interface sees
(mapper.py:257 update_map takes numpy arrays); it is never the benchmark value.

    python tools/host_boundary_rate.py [--frames 200] [--classes 40]
"""
import argparse, json, os, sys, time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--classes", type=int, default=40)
    ap.add_argument("--rule", default="height_map_log_odds")
    ap.add_argument("--chunks", type=int, default=4, help="pieces the largest arrays are staged in (1: whole)")
    ap.add_argument("--pinned", type=int, default=1, help="0: pageable copies, as the driver makes them; 1: the shim's page-locked staging buffers")
    args = ap.parse_args()
    import torch
    import semantic_search_amd as ssa
    from tests import scenes

    H, W, C = 480, 640, args.classes
    rng = np.random.default_rng(0)
    lut = scenes.schema_lut(C)
    m = ssa.SemanticMapper(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0),
                           probablistic_mapping=True, prob_aggregate=args.rule, seg_model_n_classes=C)
    m.map_seg_to_schema = lambda ids: lut[ids]
    m.pinned_staging = bool(args.pinned)
    m.staging_chunks = args.chunks
    m.reset(np.array([0.0, -0.2, 0.0, 4.5, 2.5, 4.5]))
    frames = []
    for i in range(4):
        agent, cam = scenes.make_pose(1.5 + 0.2 * i, 0.0, 1.2, 0.8 - 0.2 * i)
        fr = scenes.room_frame(rng, H, W, C, 1.0, cam, lut=lut)
        frames.append(scenes.make_obs(fr, agent, cam, goal_index=0))
    for i in range(5):
        m.update_map(dict(frames[i % 4], depth=frames[i % 4]["depth"].copy()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.frames):
        obs = frames[i % 4]
        m.update_map(dict(obs, depth=obs["depth"].copy()))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nbytes = sum(np.asarray(v).nbytes for v in frames[0].values() if isinstance(v, np.ndarray))
    print(json.dumps({"what": "SemanticMapper.update_map fed numpy observations (1 env)", "rule": args.rule,
                      "classes": C, "pinned_staging": bool(args.pinned), "chunks": args.chunks, "frames": args.frames, "frames_per_s": args.frames / dt,
                      "ms_per_frame": 1e3 * dt / args.frames, "host_bytes_per_frame": nbytes}))


if __name__ == "__main__":
    main()
