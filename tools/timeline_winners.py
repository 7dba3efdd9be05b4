"""Phase timeline of the winner kernel from a library built with -DSSMAP_TIMELINE (run on the GPU box):
    SSMAP_LIB=semantic-search_amd/lib/ab/tl.so python3 tools/timeline_winners.py [--workload W] [--envs E]
Prints, over the blocks of the last step, the mean time spent between the stamps and the spread of
the blocks' starts and ends (wall clock, 10 ns ticks)."""
import argparse, ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="log_odds_c40")
    ap.add_argument("--envs", type=int, default=16)
    a = ap.parse_args()
    import torch
    import bench
    import semantic_search_amd as ssa
    from semantic_search_amd import _capi
    from tests import scenes
    wl = bench.WORKLOADS[a.workload]
    C, span, E = wl["C"], wl["span"], a.envs
    dev = torch.device("cuda:0")
    map_hw = ssa.map_size([0, 0, 0, span, 3.0, span], 0.03)
    eng = ssa.MapEngine(E, (bench.H, bench.W), num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0),
                        probablistic_mapping=True, prob_aggregate=wl["rule"], seg_model_n_classes=C,
                        max_map_hw=map_hw, filter_goal_prob=wl.get("filter_goal_prob"), device=dev)
    for k, v in wl["attrs"].items():
        setattr(eng.params, k, v)
    eng.set_schema(scenes.schema_lut(C))
    for e in range(E):
        eng.reset_env(e, [0, 0, 0, span, 3.0, span])
    depth, logits, gt = bench.synth_frames(E, C, 0, dev)
    sem = torch.empty((E, bench.H, bench.W), dtype=torch.int32, device=dev)
    unc = torch.empty((E, bench.H, bench.W), dtype=torch.float32, device=dev)
    maxp = torch.empty((E, bench.H, bench.W), dtype=torch.float32, device=dev)
    goals = np.full(E, 3, dtype=np.int32)
    prng = np.random.default_rng(1000)
    lib = _capi.load()
    nb = 16 * E
    rows = []
    for i in range(30):
        ag, cam = bench.make_poses(E, span / 2, prng)
        eng.step(depth, ag, cam, goals, logits=logits, temperature=wl["T"], semantic=sem, semantic_unc=unc,
                 semantic_max_prob=maxp, semantic_gt=gt)
        if i >= 10:
            buf = np.zeros((nb, 12), dtype=np.uint64)
            rc = lib.ssmap_debug_timeline(buf.ctypes.data_as(ctypes.c_void_p), nb)
            assert rc == 0, rc
            rows.append(buf.astype(np.int64))
    t = np.stack(rows)[:, :, :9] * 10e-3          # microseconds
    t0 = t[:, :, 0].min(axis=1, keepdims=True)
    rel = t - t0[:, :, None]
    names = ["start skew", "first loads+clear", "fold reports", "tail prefix+barriers", "phase a (records, max)",
             "phase b (min pixel)", "ballots+prefix+compaction", "emission (+commit)", "box reduce"]
    print("mean over blocks and steps, us:")
    print(f"  {'block start after first block':34s} {rel[:, :, 0].mean():7.2f}  (max {rel[:, :, 0].max(axis=1).mean():.2f})")
    for k in range(1, 9):
        d = t[:, :, k] - t[:, :, k - 1]
        print(f"  {names[k]:34s} {d.mean():7.2f}  (max {d.max(axis=1).mean():.2f})")
    print(f"  {'block end after first start':34s} {rel[:, :, 8].mean():7.2f}  (max {rel[:, :, 8].max(axis=1).mean():.2f})")
    eng.close()


if __name__ == "__main__":
    main()
