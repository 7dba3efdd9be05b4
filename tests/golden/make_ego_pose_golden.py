"""Record the ego-map orientation the reference derives from an agent pose (build container only).

    python tests/golden/make_ego_pose_golden.py      # rewrites tests/golden/ego/ego_pose.npz

The reference's mapper.py is imported as in make_golden.py.  OpenCV is absent, so the ego maps
themselves cannot be recorded; what can be is everything up to the OpenCV calls: for a set of
agent poses `update_map` is driven once per pose and the arguments the reference hands to
cv2.getRotationMatrix2D (centre, angle in degrees: mapper.py:922-923) are captured, together
with the crop centre it computes (mapper.py:891).  The angle is np.rad2deg(phi) + 90 with phi
from the agent rotation times a half turn about y (mapper.py:291-293, 885-890); the
quaternion algebra behind it is the stand-in of thirdparty_seams.py (numpy-quaternion is
absent), so that seam stays unpinned.
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden  # noqa: E402
from tests import scenes  # noqa: E402


def main():
    ref = make_golden.load_reference()
    cv2 = sys.modules["cv2"]
    calls = []
    inert = cv2.getRotationMatrix2D

    def spy(center, angle, scale):
        calls.append((float(center[0]), float(center[1]), float(angle), float(scale)))
        return inert(center, angle, scale)

    cv2.getRotationMatrix2D = spy
    rng = np.random.default_rng(2024)
    quats = [np.array([1.0, 0.0, 0.0, 0.0]),                       # identity
             np.array([np.cos(0.4), 0.0, np.sin(0.4), 0.0]),       # yaw 0.8
             np.array([np.cos(-1.1), 0.0, np.sin(-1.1), 0.0]),
             np.array([0.0, 0.0, 1.0, 0.0]),                       # half turn about y
             np.array([2.0, 0.0, 0.5, 0.0])]                       # not normalised
    for _ in range(11):                                            # arbitrary rotations (pitch and roll too)
        q = rng.standard_normal(4)
        quats.append(q / np.linalg.norm(q))
    bounds = np.array([-1.0, -0.2, 0.5, 5.0, 2.5, 6.5])
    mapper = ref.SemanticMapper(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0))
    mapper.reset(bounds)
    poses, angles, centres, cells = [], [], [], []
    for i, q in enumerate(quats):
        pos = np.array([rng.uniform(-0.5, 4.5), 0.1, rng.uniform(1.0, 6.0)])
        agent = np.concatenate([pos, q])
        _, cam = scenes.make_pose(pos[0], 0.0, pos[2], 0.3)
        frame = scenes.uniform_frame(rng, 24, 32, 0, 1.0)
        obs = scenes.make_obs(frame, agent, cam)
        calls.clear()
        mapper.update_map(obs, None)
        assert len(calls) == 1 and calls[0][3] == 1.0
        poses.append(agent)
        angles.append(calls[0][2])
        centres.append(calls[0][:2])
        cells.append(mapper.world2map(pos, bounds[:3]))
    cv2.getRotationMatrix2D = inert
    path = os.path.join(HERE, "ego", "ego_pose.npz")
    np.savez_compressed(path, agent_pose=np.stack(poses), angle_deg=np.array(angles),
                        centre_xy=np.array(centres), agent_cell=np.stack(cells), bounds=bounds)
    print(path, "angles", np.round(angles, 3))


if __name__ == "__main__":
    main()
