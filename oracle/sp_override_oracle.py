"""CPU restatement of the shortest-path policy's found override (test infrastructure only).

Follows /root/reference/sem_objnav/obj_nav/env.py:564-605: when the mapper raises `found`, the
ground-truth map must hold the goal id inside the found disk around the robot, else the episode is
latched as a false positive; once the robot is within the success distance of the viewpoint, the
decision becomes "the ground-truth map holds the goal id at all", and a goal that is mapped but was
neither found nor mis-found is a false negative.  Pinned by tests/golden/sp_override/*.npz, which
tests/golden/make_sp_override_golden.py records by executing that block of env.py itself.
"""
from __future__ import annotations

import numpy as np

from oracle import raster
from oracle.mapper_oracle import cell_of


class SpOverride:
    def __init__(self):
        self.fp_found = False   # env.py:317
        self.fn_found = False   # env.py:318

    def questions(self, mapper, agent_pos, goal_id):
        """(goal id in gt_map inside the found disk, goal id in gt_map at all), env.py:581-593, 600."""
        robot = cell_of(np.asarray(agent_pos)[:3], mapper.map_bounds[:3], mapper.pixel2meter)
        radius = int(mapper.found_dist_m / mapper.pixel2meter)
        disk = raster.disk_mask(mapper.gt_map.shape, int(robot[0]), int(robot[1]), radius)
        return bool(np.any(mapper.gt_map[disk] == goal_id)), bool(np.any(mapper.gt_map == goal_id))

    def apply(self, true_found, in_disk, anywhere, distance_to_vp, success_distance):
        is_found = False                                     # env.py:575
        if true_found and not in_disk:                       # env.py:580-594
            self.fp_found = True
        if distance_to_vp < success_distance:                # env.py:598
            self.fn_found = bool(anywhere and not true_found and not self.fp_found)   # env.py:599-603
            if anywhere:                                     # env.py:604-605
                is_found = True
        return is_found
