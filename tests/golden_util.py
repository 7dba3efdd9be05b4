"""Loading and replaying the committed golden vectors (tests/golden/*.npz)."""
from __future__ import annotations

import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FLOAT_STATE = ("height_map", "unc_map", "goal_map", "hits", "views")
FINAL_NAMES = ("semantic_map", "semantic_map_color", "height_map", "gt_map", "unc_map", "goal_map",
               "log_odds_seg_map", "avg_prob_seg_map", "avg_prob_weight_map", "background_map",
               "hits", "views", "total_view", "cumulative_conf", "max_conf", "max_other_conf")


def case_names():
    return sorted(os.path.splitext(os.path.basename(p))[0]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


class GoldenCase:
    def __init__(self, name):
        self.data = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.meta = json.loads(str(self.data["meta"]))
        self.name = name

    @property
    def n_frames(self):
        return self.meta["n_frames"]

    def mapper_kwargs(self):
        m = self.meta
        kw = dict(num_semantic_classes=m.get("n_schema", 9), depth_hfov=m.get("hfov", 59.0),
                  depth_range=tuple(m.get("depth_range", (0.5, 5.0))),
                  probablistic_mapping=m["prob"], prob_aggregate=m["rule"],
                  seg_model_n_classes=m["C"] or None, filter_goal_prob=m["filter_goal_prob"])
        kw.update(m.get("ctor", {}))
        if m.get("collect_nb_data"):
            kw["collect_nb_data"] = True
        if m.get("nb_models"):
            kw["nb_models"] = {int(g): v for g, v in m["nb_models"].items()}
        return kw

    def goal_classes(self):
        """objnav_to_sem_pred of the case ({goal id: model classes}), or None."""
        gc = self.meta.get("goal_classes")
        return {int(g): list(v) for g, v in gc.items()} if gc else None

    def nb_rows(self, key):
        k = f"final_nb_{key}"
        return (self.data[k], self.data[k + "_chunks"]) if k in self.data.files else None

    def obs(self, i):
        pre = f"f{i}_in_"
        obs = {k[len(pre):]: self.data[k].copy() for k in self.data.files if k.startswith(pre)}
        obs["agent_position_and_rotation"] = obs.pop("agent")
        obs["camera_position_and_rotation"] = obs.pop("camera")
        obs["objectgoal"] = np.array([self.meta.get("goal_index", 0)])
        return obs

    def expect(self, i, key):
        return self.data[f"f{i}_{key}"]

    def final(self, name):
        key = "final_" + name
        return self.data[key] if key in self.data.files else None


def finite_sum(a):
    f = np.isfinite(a)
    return np.array([a[f].astype(np.float64).sum(), float(f.sum())])
