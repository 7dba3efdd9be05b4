"""Egocentric observation maps (mapper.py:876-998).

PARITY UNPINNED: the reference builds them with OpenCV, which is absent from the image, and holds
no vectors.  The CPU tests check the restated primitives (oracle/egomap_oracle.py) against
properties the published algorithms have; the GPU tests demand that the HIP kernel equals that
restatement bit for bit (uint8 colours, float32 uncertainties)."""
import math

import numpy as np
import pytest

from oracle import egomap_oracle as eo
from oracle import mapper_oracle as mo
from tests import scenes


def _yaw_quat(yaw):
    return np.array([math.cos(yaw / 2), 0.0, math.sin(yaw / 2), 0.0])      # w, x, y, z: rotation about y


def test_area_table_weights_sum_to_one():
    for ssize, dsize in ((333, 128), (500, 128), (200, 128), (129, 128)):
        tab = eo.area_table(ssize, dsize)
        sums = np.zeros(dsize)
        for d, s, w in tab:
            assert 0 <= s < ssize
            sums[d] += float(w)
        assert np.allclose(sums, 1.0, atol=1e-6)


def test_resizes_keep_constant_images_constant():
    for side in (100, 333):
        img = np.full((side, side, 3), 137, dtype=np.uint8)
        assert (eo.resize_area_u8(img, (128, 128)) == 137).all()
    unc = np.full((100, 100), 0.25, dtype=np.float32)
    assert (eo.resize_nearest(unc, (128, 128)) == np.float32(0.25)).all()


def test_integer_factor_area_resize_is_the_box_mean():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (384, 384, 3)).astype(np.uint8)
    out = eo.resize_area_u8(img, (128, 128))                  # factor 3: sum * float32(1/9), rounded half even
    blk = img[:3, :3].astype(np.int64).sum(axis=(0, 1))
    assert np.array_equal(out[0, 0], np.rint(blk.astype(np.float32) * np.float32(1 / 9)).astype(np.uint8))
    out2 = eo.resize_area_u8(img[:256, :256], (128, 128))     # factor 2: (sum + 2) >> 2
    assert np.array_equal(out2[5, 7], (img[10:12, 14:16].astype(np.int64).sum(axis=(0, 1)) + 2) >> 2)
    assert np.array_equal(eo.resize_area_u8(img[:128, :128], (128, 128)), img[:128, :128])   # factor 1


def test_enlarging_area_mode_repeats_source_pixels_between_boundaries():
    """100 -> 128: destination pixels that lie inside one source pixel copy it exactly."""
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (100, 100, 3)).astype(np.uint8)
    out = eo.resize_area_enlarge_u8(img, (128, 128))
    assert out.shape == (128, 128, 3)
    assert np.array_equal(out[0, 0], img[0, 0])
    assert np.array_equal(out[127, 127], img[99, 99])


def test_warp_by_zero_angle_is_the_identity_and_quarter_turn_rotates():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (41, 41, 3)).astype(np.uint8)
    M = eo.rotation_matrix_2d((20.5, 20.5), 0.0)
    assert np.array_equal(eo.warp_affine_nearest(img, M, (41, 41), (0, 0, 0)), img)
    M = eo.rotation_matrix_2d((20.0, 20.0), 90.0)
    out = eo.warp_affine_nearest(img, M, (41, 41), (0, 0, 0))
    assert np.array_equal(out, np.rot90(img, 1))            # positive angle = counter-clockwise


def test_heading_angle_equals_the_angle_the_reference_hands_to_opencv():
    """tests/golden/ego/ego_pose.npz holds, for 16 agent poses, the angle and centre the reference's own
    update_map -> ego_map passes to cv2.getRotationMatrix2D (mapper.py:291-293, 885-890, 922-923) and
    the crop centre (mapper.py:891), recorded by tests/golden/make_ego_pose_golden.py."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ego", "ego_pose.npz"))
    side = int(10.0 / 0.03) + 150
    for pose, angle, centre, cell in zip(g["agent_pose"], g["angle_deg"], g["centre_xy"], g["agent_cell"]):
        # numpy's vectorised arctan2 (what cartesian_to_polar calls) and libm's atan2 (the oracle, the
        # engine) differ by a few units in the last place on some inputs
        assert abs(np.rad2deg(eo.heading_angle(pose[3:])) + 90 - angle) <= 1e-12
        assert tuple(centre) == (side / 2, side / 2)
        d = pose[:3] - g["bounds"][:3]
        assert np.array_equal(np.floor(np.array([d[2], d[0]]) / 0.03).astype(np.int64), cell)
    # an agent with the identity rotation looks along -z after the half turn about y: phi = -pi/2,
    # rotation angle 0, and the ego map is the unrotated crop around the agent
    assert g["angle_deg"][0] == 0.0 and np.array_equal(g["agent_pose"][0][3:], [1.0, 0.0, 0.0, 0.0])


def test_identity_rotation_gives_the_unrotated_crop():
    rng = np.random.default_rng(5)
    palette = mo.palette(12)
    sem, color, unc = _random_maps(rng, 300, 280, palette, 3)
    pos = np.array([4.0, 0.1, 4.5])
    small, large, unc_s, unc_l = eo.ego_map(sem, color, unc, pos, np.array([1.0, 0.0, 0.0, 0.0]), np.zeros(3), 3,
                                            palette, egomap_small_side_m=3.84, egomap_large_side_m=7.68)
    # 7.68 m = 256 cells -> 2 x 2 box means of the crop centred on the agent's cell (150, 133)
    r0, c0 = 150 - 128, 133 - 128
    crop = color[r0:r0 + 256, c0:c0 + 256].astype(np.int64).copy()
    crop[sem[r0:r0 + 256, c0:c0 + 256] == 3] = palette[9]
    want = (crop.reshape(128, 2, 128, 2, 3).sum(axis=(1, 3)) + 2) >> 2
    far = np.ones((128, 128), dtype=bool)
    far[56:72, 56:72] = False                                  # away from the agent disk
    assert np.array_equal(large[far], want[far].astype(np.uint8))
    assert np.array_equal(unc_l[far], unc[r0:r0 + 256:2, c0:c0 + 256:2, 0][far])


def test_crop_pads_with_zeros_outside_the_map():
    img = np.arange(25, dtype=np.uint32).reshape(5, 5) + 1
    out = eo.crop_fn(img, np.array([0, 4]), (4, 4))
    assert out.shape == (4, 4)
    assert out[0, 0] == 0 and out[2, 0] == img[0, 2]


def _random_maps(rng, rows, cols, palette, goal):
    palette = np.asarray(palette)
    sem = rng.integers(0, 3, (rows, cols)).astype(np.uint32)
    for _ in range(25):
        r, c = rng.integers(0, rows), rng.integers(0, cols)
        sem[r:r + rng.integers(2, 15), c:c + rng.integers(2, 15)] = rng.integers(3, 9)
    color = palette[sem]
    unc = rng.random((rows, cols, 1)).astype(np.float32)
    return sem, color, unc


@pytest.mark.gpu
@pytest.mark.parametrize("prob,large_side_m,with_aux,small_side_m", [
    (True, 10.0, True, 3.0), (False, 10.0, False, 3.0), (True, 15.0, False, 3.0),
    # crops that are 1, 2, 3 and 6 times the 128-pixel output: OpenCV's integer-factor area path
    (True, 7.68, True, 3.84), (False, 11.52, False, 7.68), (True, 23.04, False, 3.0)])
def test_engine_ego_maps_equal_the_restatement(prob, large_side_m, with_aux, small_side_m):
    import torch
    import semantic_search_amd as ssa
    rng = np.random.default_rng(11)
    bounds = [np.array([0.0, -0.2, 0.0, 9.0, 2.5, 7.5]), np.array([-2.0, 0.0, 1.0, 4.0, 2.5, 13.0]),
              np.array([0.0, 0.0, 0.0, 14.39, 3.0, 14.39])]
    n = len(bounds)
    eng = ssa.MapEngine(n, (48, 64), num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0),
                        probablistic_mapping=prob, prob_aggregate="height_map_argmax", max_map_hw=(480, 480))
    eng.set_egomap(egomap_small_side_m=small_side_m, egomap_large_side_m=large_side_m)
    palette = eng.color_pallete
    host = []
    for e in range(n):
        maps = eng.reset_env(e, bounds[e])
        rows, cols = maps["semantic_map"].shape
        sem, color, unc = _random_maps(rng, rows, cols, palette, 3 + e)
        maps["semantic_map"].copy_(torch.from_numpy(sem.view(np.int32)))
        maps["semantic_map_color"].copy_(torch.from_numpy(color))
        if prob:
            maps["unc_map"].copy_(torch.from_numpy(unc))
        host.append((sem, color, unc if prob else None))
    for trial in range(3):
        poses, goals = [], []
        for e in range(n):
            b = bounds[e]
            # inside the map, at its edge, and outside of it
            fx, fz = [(0.5, 0.5), (0.02, 0.97), (-0.05, 0.4)][(trial + e) % 3]
            pos = np.array([b[0] + fx * (b[3] - b[0]), 0.1, b[2] + fz * (b[5] - b[2])])
            poses.append(np.concatenate([pos, _yaw_quat(rng.uniform(-math.pi, math.pi))]))
            goals.append(3 + (e + trial) % 6)
        aux = None
        if with_aux:
            aux = np.zeros((n, 12), dtype=np.float32)
            aux[np.arange(n), rng.integers(0, 12, n)] = 1.0
            aux[0, 5] = 0.5                                     # not one-hot: colour 127
        small, large, unc_s, unc_l = eng.ego_maps(np.stack(poses), goals, aux)
        torch.cuda.synchronize()
        for e in range(n):
            sem, color, unc = host[e]
            want = eo.ego_map(sem, color, unc, poses[e][:3], poses[e][3:], bounds[e][:3], goals[e], palette,
                              aux_angle_onehot=None if aux is None else aux[e], egomap_large_side_m=large_side_m,
                              egomap_small_side_m=small_side_m)
            assert np.array_equal(small[e].cpu().numpy(), want[0]), (trial, e, "small")
            assert np.array_equal(large[e].cpu().numpy(), want[1]), (trial, e, "large")
            if prob:
                assert np.array_equal(unc_s[e].cpu().numpy(), want[2]), (trial, e, "unc small")
                assert np.array_equal(unc_l[e].cpu().numpy(), want[3]), (trial, e, "unc large")
            else:
                assert unc_s is None and want[2] is None
    eng.close()


@pytest.mark.gpu
def test_mapper_update_map_returns_ego_maps():
    """The drop-in class returns the four ego arrays with the reference's shapes and dtypes and
    they show the freshly fused map (not blank)."""
    import semantic_search_amd as ssa
    rng = np.random.default_rng(2)
    lut = scenes.small_lut(8)
    m = ssa.SemanticMapper(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0),
                           probablistic_mapping=True, prob_aggregate="height_map_argmax", seg_model_n_classes=8)
    m.map_seg_to_schema = lambda ids: lut[ids]
    m.reset(np.array([0.0, -0.2, 0.0, 4.5, 2.5, 4.5]))
    agent, cam = scenes.make_pose(1.5, 0.0, 1.2, 0.8)
    frame = scenes.room_frame(rng, 96, 128, 8, 1.0, cam, lut=lut)
    ego, found = m.update_map(scenes.make_obs(frame, agent, cam, goal_index=0), np.eye(12, dtype=np.float32)[3])
    small, large, unc_s, unc_l = ego
    assert small.shape == (128, 128, 3) and small.dtype == np.uint8 and large.shape == (128, 128, 3)
    assert unc_s.shape == (128, 128) and unc_s.dtype == np.float32 and unc_l.shape == (128, 128)
    assert small.any() and large.any()
    want = eo.ego_map(m.semantic_map, m.semantic_map_color, m.unc_map, agent[:3], agent[3:], m.map_bounds[:3],
                      int(m.goal2sem_id[0]), m.color_pallete, aux_angle_onehot=np.eye(12, dtype=np.float32)[3])
    assert np.array_equal(small, want[0]) and np.array_equal(large, want[1])
    assert np.array_equal(unc_s, want[2]) and np.array_equal(unc_l, want[3])


@pytest.mark.gpu
def test_engine_ego_maps_more_envs_than_one_launch_takes():
    """18 envs: the poses travel in packs of 16 per launch; every env still gets its own maps."""
    import torch
    import semantic_search_amd as ssa
    rng = np.random.default_rng(12)
    n = 18
    bounds = [np.array([0.0, 0.0, 0.0, 6.0 + 0.3 * (e % 4), 2.5, 5.0 + 0.2 * (e % 5)]) for e in range(n)]
    eng = ssa.MapEngine(n, (48, 64), num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0),
                        probablistic_mapping=True, prob_aggregate="height_map_argmax", max_map_hw=(240, 240))
    eng.set_egomap()
    palette = eng.color_pallete
    host, poses, goals = [], [], []
    for e in range(n):
        maps = eng.reset_env(e, bounds[e])
        rows, cols = maps["semantic_map"].shape
        sem, color, unc = _random_maps(rng, rows, cols, palette, 3)
        maps["semantic_map"].copy_(torch.from_numpy(sem.view(np.int32)))
        maps["semantic_map_color"].copy_(torch.from_numpy(color))
        maps["unc_map"].copy_(torch.from_numpy(unc))
        host.append((sem, color, unc))
        pos = np.array([rng.uniform(0.5, 5.5), 0.1, rng.uniform(0.5, 4.5)])
        poses.append(np.concatenate([pos, _yaw_quat(rng.uniform(-math.pi, math.pi))]))
        goals.append(3 + e % 6)
    small, large, unc_s, unc_l = eng.ego_maps(np.stack(poses), goals, None)
    torch.cuda.synchronize()
    for e in (0, 7, 15, 16, 17):
        sem, color, unc = host[e]
        want = eo.ego_map(sem, color, unc, poses[e][:3], poses[e][3:], bounds[e][:3], goals[e], palette)
        assert np.array_equal(small[e].cpu().numpy(), want[0]) and np.array_equal(large[e].cpu().numpy(), want[1]), e
        assert np.array_equal(unc_s[e].cpu().numpy(), want[2]) and np.array_equal(unc_l[e].cpu().numpy(), want[3]), e
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(24))
def test_engine_ego_maps_random_configurations(seed):
    """Random crop sizes (shrinking, enlarging and integer factors), output sizes (square or not),
    agent radii, auxiliary bins, map sizes smaller and larger than the crops, poses inside and
    outside of the map: the kernel equals the restatement bit for bit."""
    import torch
    import semantic_search_amd as ssa
    rng = np.random.default_rng(900 + seed)
    prob = bool(seed % 2)
    small_m = float(rng.choice([1.5, 2.0, 3.0, 3.84, 4.5]))
    large_m = float(rng.choice([6.0, 7.68, 10.0, 12.5, 16.0]))
    size = [(128, 128), (64, 64), (96, 128), (100, 100), (128, 80)][seed % 5]
    radius_m = float(rng.choice([0.06, 0.1, 0.25]))
    bins = int(rng.choice([8, 12]))
    offset = int(rng.choice([0, 60, 150]))
    cfg = dict(egomap_small_side_m=small_m, egomap_large_side_m=large_m, egomap_size=size,
               agent_draw_radius_m=radius_m, aux_num_bins=bins, offset_for_cut=offset)
    bounds = np.array([rng.uniform(-2, 1), 0.0, rng.uniform(-2, 1), 0, 2.5, 0])
    bounds[3] = bounds[0] + rng.uniform(2.0, 13.0)
    bounds[5] = bounds[2] + rng.uniform(2.0, 13.0)
    eng = ssa.MapEngine(1, (48, 64), num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0),
                        probablistic_mapping=prob, prob_aggregate="height_map_argmax", max_map_hw=(480, 480))
    eng.set_egomap(**cfg)
    palette = eng.color_pallete
    maps = eng.reset_env(0, bounds)
    rows, cols = maps["semantic_map"].shape
    sem, color, unc = _random_maps(rng, rows, cols, palette, 3)
    maps["semantic_map"].copy_(torch.from_numpy(sem.view(np.int32)))
    maps["semantic_map_color"].copy_(torch.from_numpy(color))
    if prob:
        maps["unc_map"].copy_(torch.from_numpy(unc))
    for trial in range(3):
        fx, fz = rng.uniform(-0.15, 1.15, 2)
        pos = np.array([bounds[0] + fx * (bounds[3] - bounds[0]), 0.1, bounds[2] + fz * (bounds[5] - bounds[2])])
        pose = np.concatenate([pos, _yaw_quat(rng.uniform(-math.pi, math.pi))])
        goal = int(rng.integers(3, 9))
        aux = None
        if trial != 1:
            aux = np.zeros((1, bins), dtype=np.float32)
            aux[0, rng.integers(0, bins)] = 1.0
        small, large, unc_s, unc_l = eng.ego_maps(pose[None], [goal], aux)
        torch.cuda.synchronize()
        want = eo.ego_map(sem, color, unc if prob else None, pose[:3], pose[3:], bounds[:3], goal, palette,
                          aux_angle_onehot=None if aux is None else aux[0], **cfg)
        assert np.array_equal(small[0].cpu().numpy(), want[0]), (trial, "small", cfg)
        assert np.array_equal(large[0].cpu().numpy(), want[1]), (trial, "large", cfg)
        if prob:
            assert np.array_equal(unc_s[0].cpu().numpy(), want[2]), (trial, "unc small", cfg)
            assert np.array_equal(unc_l[0].cpu().numpy(), want[3]), (trial, "unc large", cfg)
    eng.close()
