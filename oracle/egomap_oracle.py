"""TEST INFRASTRUCTURE ONLY -- CPU restatement of `SemanticMapper.ego_map`
(sem_objnav/obj_nav/mapper.py:876-998 with crop_fn :1000-1008, pad_img_to_fit_bbox :1010-1025,
draw_aux_angles :1027-1039, draw_agent :1041-1051).

Only tests/ may import this module.  PARITY UNPINNED: the method is built on OpenCV
(`opencv-python==4.7.0.72`, environment.yaml:232), which is absent from the image, and the
reference holds no vectors for it.  The OpenCV primitives are restated here from their
published 4.x algorithms:
  * getRotationMatrix2D: alpha = cos(a), beta = sin(a), [[alpha, beta, (1-alpha)cx - beta cy],
    [-beta, alpha, beta cx + (1-alpha)cy]] in float64.
  * warpAffine, INTER_NEAREST, BORDER_CONSTANT: the matrix is inverted in float64, then the
    source coordinates are 22.10 fixed point: X = (round(M00 x 1024) + round((M01 y + M02) 1024)
    + 512) >> 10 (round half to even), same for Y; out-of-range reads give the border value.
  * resize, INTER_AREA, shrinking by a non-integer factor: separable area table (whole source
    cells weigh 1/cell, the two partial cells their covered fraction), float32 accumulation,
    horizontal pass first, then the rows of one output row in order; round half to even.
  * resize, INTER_AREA, enlarging: the bilinear kernel with the area-mode coefficients
    fx = (dx+1) - (sx+1)/scale, 11-bit fixed-point weights and the 8-bit vertical pass
    ((b0 (S0 >> 4)) >> 16) + ((b1 (S1 >> 4)) >> 16) + 2 >> 2.
  * resize, INTER_NEAREST: sx = min(floor(dx * src/dst), src - 1).
  * circle, filled: oracle/raster.py.
"""
from __future__ import annotations

import math

import numpy as np

from oracle import raster
from oracle.mapper_oracle import quat_from_rotvec, quat_mul, quat_to_matrix

AB_BITS = 10
AB_SCALE = 1 << AB_BITS
COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


# ----------------------------------------------------------------------------- numpy helpers of the reference
def crop_fn(img, center, output_size):
    """mapper.py:1000-1025 (zero padding when the window leaves the image)."""
    h, w = np.array(output_size, dtype=int)
    x = int(center[0] - h / 2)
    y = int(center[1] - w / 2)
    x1, x2, y1, y2 = x, x + h, y, y + w
    if x1 < 0 or y1 < 0 or x2 > img.shape[0] or y2 > img.shape[1]:
        left, right = abs(min(0, x1)), max(x2 - img.shape[0], 0)
        top, bottom = abs(min(0, y1)), max(y2 - img.shape[1], 0)
        pad = ((left, right), (top, bottom)) + (((0, 0),) if img.ndim == 3 else ())
        img = np.pad(img, pad, mode="constant")
        x1, x2, y1, y2 = x1 + left, x2 + left, y1 + top, y2 + top
    return img[x1:x2, y1:y2]


def aux_points(aux_num_bins=12, aux_radius_m=0.4, pixel2meter=0.03):
    """mapper.py:78-89."""
    bin_size = 360 / aux_num_bins
    radius_px = int(aux_radius_m / pixel2meter)
    pts = [-np.array((radius_px * np.cos(np.deg2rad(d)), radius_px * np.sin(np.deg2rad(d))))
           for d in np.arange(0, 360, bin_size)]
    return np.array(pts, dtype=np.int32)


# ----------------------------------------------------------------------------- OpenCV restatements
def rotation_matrix_2d(center_xy, angle_deg, scale=1.0):
    a = angle_deg * (math.pi / 180.0)          # angle *= CV_PI / 180
    alpha, beta = math.cos(a) * scale, math.sin(a) * scale
    cx, cy = float(np.float32(center_xy[0])), float(np.float32(center_xy[1]))   # Point2f
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy],
                     [-beta, alpha, beta * cx + (1 - alpha) * cy]], dtype=np.float64)


def invert_affine(M):
    m = [float(v) for v in np.asarray(M, dtype=np.float64).reshape(6)]
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11
    m[1] *= -D
    m[3] *= -D
    m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def _sat_int(v):
    """saturate_cast<int>(double): round half to even, clamp to int32."""
    return np.clip(np.rint(v), -2147483648, 2147483647).astype(np.int64)


def warp_source_coords(M, dsize_wh):
    """Source (X, Y) of every destination pixel for the nearest-neighbour warp."""
    m = invert_affine(M)
    w, h = dsize_wh
    xs = np.arange(w, dtype=np.float64)
    ys = np.arange(h, dtype=np.float64)
    adelta = _sat_int(m[0] * xs * AB_SCALE)
    bdelta = _sat_int(m[3] * xs * AB_SCALE)
    X0 = _sat_int((m[1] * ys + m[2]) * AB_SCALE) + AB_SCALE // 2
    Y0 = _sat_int((m[4] * ys + m[5]) * AB_SCALE) + AB_SCALE // 2
    X = (X0[:, None] + adelta[None, :]) >> AB_BITS
    Y = (Y0[:, None] + bdelta[None, :]) >> AB_BITS
    return np.clip(X, -32768, 32767), np.clip(Y, -32768, 32767)     # saturate_cast<short>


def warp_affine_nearest(src, M, dsize_wh, border):
    X, Y = warp_source_coords(M, dsize_wh)
    inside = (X >= 0) & (X < src.shape[1]) & (Y >= 0) & (Y < src.shape[0])
    out = np.empty((dsize_wh[1], dsize_wh[0]) + src.shape[2:], dtype=src.dtype)
    out[...] = border
    out[inside] = src[Y[inside], X[inside]]
    return out


def area_table(ssize, dsize):
    """computeResizeAreaTab for one axis: list of (dst index, src index, float32 weight)."""
    scale, _ = axis_scale(ssize, dsize)
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def resize_area_shrink_u8(src, dsize_wh):
    """INTER_AREA, both scale factors > 1 and not both integer."""
    sh, sw = src.shape[:2]
    dw, dh = dsize_wh
    cn = src.shape[2] if src.ndim == 3 else 1
    S = src.reshape(sh, sw, cn)
    xtab, ytab = area_table(sw, dw), area_table(sh, dh)
    out = np.zeros((dh, dw, cn), dtype=np.uint8)
    acc = np.zeros((dw, cn), dtype=np.float32)
    prev_dy = ytab[0][0]
    for dy, sy, beta in ytab:
        buf = np.zeros((dw, cn), dtype=np.float32)
        row = S[sy].astype(np.float32)
        for dx, sx, alpha in xtab:                       # float32, in table order
            buf[dx] = buf[dx] + row[sx] * alpha
        if dy != prev_dy:
            out[prev_dy] = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
            acc = beta * buf
            prev_dy = dy
        else:
            acc = acc + beta * buf
    out[prev_dy] = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
    return out.reshape((dh, dw) + src.shape[2:])


def resize_area_integer_u8(src, dsize_wh):
    """INTER_AREA with integer factors: plain box mean, (sum * (1/area)) in float32, rounded;
    the factor 2 x 2 is rounded half up in integers, (sum + 2) >> 2; factor 1 is a copy."""
    sh, sw = src.shape[:2]
    dw, dh = dsize_wh
    fx, fy = sw // dw, sh // dh
    cn = src.shape[2] if src.ndim == 3 else 1
    S = src.reshape(sh, sw, cn).astype(np.int64)
    sums = S.reshape(dh, fy, dw, fx, cn).sum(axis=(1, 3))
    if fx == 2 and fy == 2:
        return ((sums + 2) >> 2).astype(np.uint8).reshape((dh, dw) + src.shape[2:])
    out = np.clip(np.rint(sums.astype(np.float32) * np.float32(1.0 / (fx * fy))), 0, 255).astype(np.uint8)
    return out.reshape((dh, dw) + src.shape[2:])


def axis_scale(ssize, dsize):
    """(scale, inv_scale) the way resize() derives them: inv = dst/src, scale = 1/inv."""
    inv = dsize / ssize
    return 1.0 / inv, inv


def _linear_area_mode_axis(ssize, dsize, horizontal):
    """Source index and 11-bit weights per destination index, area-mode bilinear.  Only the
    horizontal table pins the weight to the last source column; vertically the second row is
    clipped instead."""
    scale, inv = axis_scale(ssize, dsize)
    ofs, w0, w1 = [], [], []
    for dx in range(dsize):
        sx = math.floor(dx * scale)
        fx = np.float32((dx + 1) - (sx + 1) * inv)
        fx = np.float32(0.0) if fx <= 0 else np.float32(fx - np.floor(fx))
        if horizontal and sx < 0:
            fx, sx = np.float32(0.0), 0
        if horizontal and sx >= ssize - 1:
            fx, sx = np.float32(0.0), ssize - 1
        ofs.append(sx)
        w0.append(int(np.clip(np.rint(np.float32(1.0 - fx) * np.float32(COEF_SCALE)), -32768, 32767)))
        w1.append(int(np.clip(np.rint(fx * np.float32(COEF_SCALE)), -32768, 32767)))
    return np.array(ofs), np.array(w0, dtype=np.int64), np.array(w1, dtype=np.int64)


def resize_area_enlarge_u8(src, dsize_wh):
    """INTER_AREA when the image grows: bilinear kernel, area-mode coefficients, 8-bit fixed point."""
    sh, sw = src.shape[:2]
    dw, dh = dsize_wh
    cn = src.shape[2] if src.ndim == 3 else 1
    S = src.reshape(sh, sw, cn).astype(np.int64)
    xo, xa0, xa1 = _linear_area_mode_axis(sw, dw, True)
    yo, yb0, yb1 = _linear_area_mode_axis(sh, dh, False)
    x1 = np.minimum(xo + 1, sw - 1)
    y1 = np.minimum(yo + 1, sh - 1)
    # horizontal pass on the two source rows of every output row (int32 intermediate)
    H = S[:, xo, :] * xa0[None, :, None] + S[:, x1, :] * xa1[None, :, None]
    r0, r1 = H[yo], H[y1]
    out = (((yb0[:, None, None] * (r0 >> 4)) >> 16) + ((yb1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8).reshape((dh, dw) + src.shape[2:])


def resize_area_u8(src, dsize_wh):
    sh, sw = src.shape[:2]
    dw, dh = dsize_wh
    sx, sy = axis_scale(sw, dw)[0], axis_scale(sh, dh)[0]
    if sx >= 1 and sy >= 1:
        if abs(sx - int(sx)) < 2.220446049250313e-16 and abs(sy - int(sy)) < 2.220446049250313e-16:
            return resize_area_integer_u8(src, dsize_wh)
        return resize_area_shrink_u8(src, dsize_wh)
    return resize_area_enlarge_u8(src, dsize_wh)


def resize_nearest(src, dsize_wh):
    sh, sw = src.shape[:2]
    dw, dh = dsize_wh
    xs = np.minimum(np.floor(np.arange(dw) * axis_scale(sw, dw)[0]).astype(np.int64), sw - 1)
    ys = np.minimum(np.floor(np.arange(dh) * axis_scale(sh, dh)[0]).astype(np.int64), sh - 1)
    return src[ys][:, xs]


# ----------------------------------------------------------------------------- the method
def heading_angle(agent_quat_wxyz):
    """phi of mapper.py:885-890: polar angle, in the x/z plane, of (0, 0, 1) rotated by the agent
    rotation update_map hands to ego_map, which is the pose quaternion times a half turn about y
    (mapper.py:291-293, 840-847)."""
    q = quat_mul(np.asarray(agent_quat_wxyz, dtype=np.float64), quat_from_rotvec(0.0, math.pi, 0.0))
    R = quat_to_matrix(q)
    heading = R @ np.array([0.0, 0.0, 1.0])
    return math.atan2(heading[2], heading[0])


def ego_map(semantic_map, semantic_map_color, unc_map, agent_position, agent_quat_wxyz, lower_bound_m,
            goal_id, palette, num_semantic_classes=9, aux_angle_onehot=None, pixel2meter=0.03,
            egomap_small_side_m=3.0, egomap_large_side_m=10.0, egomap_size=(128, 128),
            agent_draw_radius_m=0.1, aux_num_bins=12, aux_radius_m=0.4, aux_bin_m=0.15, offset_for_cut=150):
    """Returns (ego_small, ego_large, unc_small, unc_large); unc_* are None when unc_map is None."""
    small = int(egomap_small_side_m / pixel2meter)
    large = int(egomap_large_side_m / pixel2meter)
    side = large + offset_for_cut
    active_goal_color = palette[num_semantic_classes]
    arrow_color = palette[num_semantic_classes + 2]
    phi = heading_angle(agent_quat_wxyz)
    d = np.asarray(agent_position, dtype=np.float64) - np.asarray(lower_bound_m, dtype=np.float64)
    pos = np.floor(np.array([d[2], d[0]]) / pixel2meter).astype(np.int64)         # mapper.py:891

    cropped = crop_fn(semantic_map, pos, (side, side))                            # mapper.py:904-911
    cropped_color = crop_fn(semantic_map_color, pos, (side, side)).copy()
    cropped_color[cropped == goal_id] = active_goal_color                          # mapper.py:920
    w, h, _ = cropped_color.shape
    M = rotation_matrix_2d((h / 2, w / 2), np.rad2deg(phi) + 90, 1.0)              # mapper.py:922-923
    ego = warp_affine_nearest(cropped_color, M, (h, w), (0, 0, 0))                 # mapper.py:924-931
    origin = np.array([ego.shape[0] / 2, ego.shape[1] / 2], dtype=np.float32).astype(np.int32)
    disk = raster.disk_mask(ego.shape[:2], int(origin[1]), int(origin[0]), int(agent_draw_radius_m / pixel2meter))
    ego[disk] = arrow_color                                                        # mapper.py:1041-1051 (x, y) = origin
    if aux_angle_onehot is not None:                                               # mapper.py:1027-1039
        org = np.array([ego.shape[0] / 2, ego.shape[1] / 2], dtype=np.int32)
        half = int(aux_bin_m / pixel2meter) // 2
        for j, p in enumerate(aux_points(aux_num_bins, aux_radius_m, pixel2meter)):
            p = p + org
            ego[int(p[0]) - half:int(p[0]) + half, int(p[1]) - half:int(p[1]) + half, :] = \
                (np.array([255, 255, 255], dtype=np.uint8) * aux_angle_onehot[j]).astype(np.uint8)
    centre = (ego.shape[0] / 2, ego.shape[1] / 2)
    ego_small = resize_area_u8(crop_fn(ego, centre, (small, small)), egomap_size)  # mapper.py:937-956
    ego_large = resize_area_u8(crop_fn(ego, centre, (large, large)), egomap_size)
    if unc_map is None:
        return ego_small, ego_large, None, None
    cropped_unc = crop_fn(unc_map, pos, (side, side))                              # mapper.py:959-966
    ego_unc = warp_affine_nearest(cropped_unc.reshape(side, side), M, (h, w), np.float32(-0.5))
    centre = (ego_unc.shape[0] / 2, ego_unc.shape[1] / 2)
    unc_small = resize_nearest(crop_fn(ego_unc, centre, (small, small)), egomap_size)
    unc_large = resize_nearest(crop_fn(ego_unc, centre, (large, large)), egomap_size)
    return ego_small, ego_large, unc_small, unc_large
