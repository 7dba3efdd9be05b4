"""Record golden vectors of the shortest-path-policy found override (build container only).

    python tests/golden/make_sp_override_golden.py     # rewrites tests/golden/sp_override/*.npz

The override is the block `if self._sp_follower is not None:` of MappingObjNavEnv.step
(/root/reference/sem_objnav/obj_nav/env.py:564-605).  env.py cannot be imported here (habitat,
gymnasium and stable-baselines are absent), so this script reads that block from the file at run
time, and executes it as it stands against a stand-in `self` whose `mapping` is the reference's own
SemanticMapper (imported like make_golden.py does) after each frame of a golden case; the simulator
calls of the block (geodesic distance, success distance) return the values of a schedule.  Nothing
of the reference is written anywhere: the outputs are the three flags the block sets, frame by
frame, next to the schedule that was fed in.  cv2.circle is the stand-in of thirdparty_seams.py
(raster parity unpinned, as in the other fixtures).
"""
from __future__ import annotations

import json
import os
import sys
import textwrap
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden  # noqa: E402
from tests import scenes  # noqa: E402

REF_ENV = "/root/reference/sem_objnav/obj_nav/env.py"
OUT = os.path.join(HERE, "sp_override")

# case of make_golden.py -> per-frame (distance to the viewpoint, success distance); the found
# distance of the mapper is varied too (a disk that misses the goal cells makes a false positive)
SCHEDULES = {
    "avg_prob_c8_walk": dict(dist=[3.0, 2.0, 0.9, 0.05, 0.05], success=0.1),
    "argmax_filter": dict(dist=[0.05, 3.0, 3.0, 0.05], success=0.1),
    "goal_decay_filter": dict(dist=[2.0, 2.0, 0.05, 1.0, 0.08, 0.02], success=0.1),
    "log_odds_c8": dict(dist=[1.0, 1.0, 1.0, 1.0, 0.01], success=0.2, found_dist_m=0.2),
    "hv_gate_filter": dict(dist=[0.3, 0.3, 0.3, 0.3, 0.3, 0.05], success=0.1, found_dist_m=0.35),
    "uniform_goal_decay": dict(dist=[0.05, 1.0, 0.05], success=0.1, goal_index=2),
    "negative_wrap_avg": dict(dist=[0.05, 0.05, 0.05], success=0.1),
    "schema41_hv": dict(dist=[1.0, 0.05, 0.05, 1.0], success=0.1, found_dist_m=0.4),
    # without probabilistic mapping gt_map is the semantic map itself (mapper.py:257)
    "direct_height_argmax": dict(dist=[0.05, 1.0, 0.05], success=0.1),
}


def override_block():
    """The source lines of the override, found by its opening test inside `def step`."""
    lines = open(REF_ENV).read().split("\n")
    start = next(i for i, ln in enumerate(lines) if ln.strip() == "if self._sp_follower is not None:"
                 and any("def step(" in p for p in lines[max(0, i - 80):i]))
    indent = len(lines[start]) - len(lines[start].lstrip())
    end = start + 1
    while end < len(lines) and (not lines[end].strip() or len(lines[end]) - len(lines[end].lstrip()) > indent):
        end += 1
    return compile(textwrap.dedent("\n".join(lines[start:end])), REF_ENV, "exec"), (start + 1, end)


class _Sim:
    def __init__(self):
        self.distance = 0.0

    def geodesic_distance(self, a, b):
        return self.distance


def run_case(ref, block, case, sched):
    frames, lut = make_golden.build_frames(case)
    mapper = ref.SemanticMapper(
        num_semantic_classes=case.get("n_schema", 9), depth_hfov=case.get("hfov", 59.0),
        depth_range=tuple(case.get("depth_range", (0.5, 5.0))), probablistic_mapping=case["prob"],
        prob_aggregate=case["rule"], seg_model_n_classes=case["C"] or None,
        filter_goal_prob=case.get("filter_goal_prob"), **case.get("ctor", {}))
    for k, v in case.get("attrs", {}).items():
        setattr(mapper, k, v)
    if "found_dist_m" in sched:
        mapper.found_dist_m = sched["found_dist_m"]
    if lut is not None:
        mapper.map_seg_to_schema = lambda ids: lut[ids]
    mapper.reset(np.array(make_golden.BOUNDS, dtype=np.float64))
    goal_index = sched.get("goal_index", case.get("goal_index", 0))

    sim = _Sim()
    success = types.SimpleNamespace(_success_distance=sched["success"])
    env = types.SimpleNamespace(habitat_env=types.SimpleNamespace(
        sim=sim, task=types.SimpleNamespace(measurements=types.SimpleNamespace(measures={"success": success}))))
    self = types.SimpleNamespace(_sp_follower=object(), _env=env, _sp_viewpoint_pos=np.zeros(3), mapping=mapper,
                                 _is_found=None, _sp_fp_found=False, _sp_fn_found=False)
    import cv2  # the stand-in installed by thirdparty_seams
    rec = dict(true_found=[], is_found=[], fp=[], fn=[], in_disk=[], anywhere=[])
    for i, (fr, agent, cam) in enumerate(frames):
        obs = scenes.make_obs(fr, agent, cam, goal_index=goal_index)
        _, found = mapper.update_map(obs, None)
        self._is_found = found
        sim.distance = sched["dist"][i]
        exec(block, dict(self=self, obs=obs, agent_pos=obs["agent_position_and_rotation"][:3], np=np, cv2=cv2))
        rec["true_found"].append(bool(found))
        rec["is_found"].append(bool(self._is_found))
        rec["fp"].append(bool(self._sp_fp_found))
        rec["fn"].append(bool(self._sp_fn_found))
        # what the block looked at, for the engine's two map queries (computed here from the
        # reference mapper's arrays with the same stand-in raster)
        goal_id = mapper.goal2sem_id[obs["objectgoal"]][0]
        px = mapper.world2map(obs["agent_position_and_rotation"][:3], mapper.map_bounds[:3])
        disk = cv2.circle(np.zeros_like(mapper.semantic_map, dtype=np.uint8), tuple(px[::-1]),
                          int(mapper.found_dist_m / mapper.pixel2meter), 1, -1).astype(bool)
        rec["in_disk"].append(bool(np.any(mapper.gt_map[disk] == goal_id)))
        rec["anywhere"].append(bool(np.any(mapper.gt_map == goal_id)))
    out = {k: np.array(v, dtype=np.int8) for k, v in rec.items()}
    out["dist"] = np.array(sched["dist"], dtype=np.float64)
    out["meta"] = np.array(json.dumps(dict(case=case["name"], success=sched["success"], goal_index=goal_index,
                                           found_dist_m=sched.get("found_dist_m"))))
    return out


def main():
    ref = make_golden.load_reference()
    block, span = override_block()
    print(f"override block: env.py lines {span[0]}-{span[1]}")
    os.makedirs(OUT, exist_ok=True)
    cases = {c["name"]: c for c in make_golden.CASES}
    for name, sched in SCHEDULES.items():
        out = run_case(ref, block, cases[name], sched)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(name, "true", out["true_found"].tolist(), "found", out["is_found"].tolist(),
              "fp", out["fp"].tolist(), "fn", out["fn"].tolist(), "disk", out["in_disk"].tolist(),
              "any", out["anywhere"].tolist())


if __name__ == "__main__":
    main()
