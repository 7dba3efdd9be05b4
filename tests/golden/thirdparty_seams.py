"""Stand-in modules for the third-party libraries the reference mapper imports but
this image lacks (numpy_indexed, numpy-quaternion, opencv, habitat.tasks.utils).

Used ONLY by make_golden.py, in the build container, to import the reference's
`sem_objnav/obj_nav/mapper.py` and record golden vectors from it.  Nothing here is
shipped or imported by the product path.  Each function restates the published
behaviour of the pinned library version (reference `environment.yaml`):

  numpy-indexed 0.3.7      group_by(keys).argmax(values): per-group maximum, first
                           occurrence of the maximum in each group.
  numpy-quaternion 2022.4  Hamilton product, exp-map from a rotation vector,
                           as_rotation_matrix (array branch, divides by the Cayley
                           norm), rotate_vectors = tensordot(matrix, vectors).
  opencv-python 4.7.0.72   filled `circle` (midpoint algorithm of drawing.cpp),
                           `erode` with a rectangular kernel and +inf border.
  habitat-lab              cartesian_to_polar (habitat/tasks/utils.py:39-42).

Parity at these seams is therefore UNPINNED by the golden vectors: they record the
reference's own code running on top of these restatements.  `cv2.line`,
`warpAffine`, `resize` and `getRotationMatrix2D` are inert (blank ego maps) because their
rasterisation cannot be restated reliably from memory; `cv2.line` draws an approximate thick
line (see `line` below), so the path overlay of mapper.py:822-833 takes part in the fixtures.
"""
from __future__ import annotations

import math
import sys
import types

import numpy as np


# ----------------------------------------------------------------------------- numpy_indexed
class _GroupBy:
    def __init__(self, keys):
        keys = np.asarray(keys)
        if keys.ndim == 1:
            keys = keys[:, None]
        self._unique, self._inverse = np.unique(keys, axis=0, return_inverse=True)
        self._inverse = self._inverse.reshape(-1)

    def argmax(self, values):
        values = np.asarray(values)
        n = values.shape[0]
        # primary: group; then value descending; then original index ascending
        order = np.lexsort((np.arange(n), -values, self._inverse))
        grp_sorted = self._inverse[order]
        first = np.ones(n, dtype=bool)
        first[1:] = grp_sorted[1:] != grp_sorted[:-1]
        return self._unique, order[first]


def _make_numpy_indexed():
    m = types.ModuleType("numpy_indexed")
    m.group_by = _GroupBy
    return m


# ----------------------------------------------------------------------------- quaternion
class Quat:
    __slots__ = ("w", "x", "y", "z")

    def __init__(self, w, x, y, z):
        self.w, self.x, self.y, self.z = float(w), float(x), float(y), float(z)

    def __mul__(self, o):
        a, b = self, o
        return Quat(
            a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
            a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
            a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w,
        )

    def norm(self):  # Cayley norm: sum of squares
        return self.w * self.w + self.x * self.x + self.y * self.y + self.z * self.z


def from_float_array(a):
    a = np.asarray(a, dtype=np.float64)
    return Quat(a[0], a[1], a[2], a[3])


def from_rotation_vector(rot):
    rot = np.asarray(rot, dtype=np.float64)
    x, y, z = rot[0] / 2, rot[1] / 2, rot[2] / 2
    vnorm = math.sqrt(x * x + y * y + z * z)
    if vnorm > 1e-14:  # quaternion exp of a pure-vector quaternion
        s = math.sin(vnorm) / vnorm
        e = math.exp(0.0)
        return Quat(e * math.cos(vnorm), e * s * x, e * s * y, e * s * z)
    return Quat(math.exp(0.0), 0.0, 0.0, 0.0)


def as_rotation_matrix(q: Quat):
    n = q.norm()
    if n == 0.0:
        raise ZeroDivisionError("quaternion has zero norm")
    m = np.empty((3, 3))
    m[0, 0] = 1.0 - 2 * (q.y ** 2 + q.z ** 2) / n
    m[0, 1] = 2 * (q.x * q.y - q.z * q.w) / n
    m[0, 2] = 2 * (q.x * q.z + q.y * q.w) / n
    m[1, 0] = 2 * (q.x * q.y + q.z * q.w) / n
    m[1, 1] = 1.0 - 2 * (q.x ** 2 + q.z ** 2) / n
    m[1, 2] = 2 * (q.y * q.z - q.x * q.w) / n
    m[2, 0] = 2 * (q.x * q.z - q.y * q.w) / n
    m[2, 1] = 2 * (q.y * q.z + q.x * q.w) / n
    m[2, 2] = 1.0 - 2 * (q.x ** 2 + q.y ** 2) / n
    return m


def rotate_vectors(R: Quat, v, axis=-1):
    v = np.asarray(v, dtype=float)
    m = as_rotation_matrix(R)
    out = np.tensordot(m, v, axes=(-1, axis))
    return np.moveaxis(out, 0, axis % v.ndim)


def _make_quaternion():
    m = types.ModuleType("quaternion")
    m.quaternion = Quat
    m.from_float_array = from_float_array
    m.from_rotation_vector = from_rotation_vector
    m.as_rotation_matrix = as_rotation_matrix
    m.rotate_vectors = rotate_vectors
    return m


# ----------------------------------------------------------------------------- cv2
def circle_halfwidths(radius: int) -> np.ndarray:
    """Half-width of the filled midpoint circle on each row offset 0..radius."""
    hw = np.full(radius + 1, -1, dtype=np.int64)
    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1
    while dx >= dy:
        hw[dy] = max(hw[dy], dx)
        hw[dx] = max(hw[dx], dy)
        dy += 1
        err += plus
        plus += 2
        mask = (1 if err <= 0 else 0) - 1  # 0 or -1
        err -= minus & mask
        dx += mask
        minus -= mask & 2
    return hw


def circle(img, center, radius, color, thickness=1, *a, **k):
    assert thickness < 0, "only filled circles are restated"
    cx, cy = int(center[0]), int(center[1])
    hw = circle_halfwidths(int(radius))
    h, w = img.shape[:2]
    for off in range(-int(radius), int(radius) + 1):
        y = cy + off
        if y < 0 or y >= h:
            continue
        half = hw[abs(off)]
        if half < 0:
            continue
        x0, x1 = max(cx - half, 0), min(cx + half, w - 1)
        if x0 <= x1:
            img[y, x0 : x1 + 1] = color
    return img


def line(img, p0, p1, color, thickness=1, *a, **k):
    """Stand-in for cv2.line: centre line of OpenCV's LineIterator under a square brush of
    `thickness` cells (oracle/raster.py line_cells).  OpenCV's thick-line polygon is NOT restated:
    the fixtures pin how the mapper uses the path mask, not the raster of the line itself."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from oracle import raster
    return raster.line(img, p0, p1, color, thickness)


def erode(src, kernel, iterations=1, **k):
    assert iterations == 1
    kh, kw = kernel.shape
    ay, ax = kh // 2, kw // 2
    h, w = src.shape
    big = np.iinfo(src.dtype).max if np.issubdtype(src.dtype, np.integer) else np.inf
    pad = np.full((h + kh, w + kw), big, dtype=src.dtype)
    pad[ay : ay + h, ax : ax + w] = src
    out = np.full((h, w), big, dtype=src.dtype)
    for i in range(kh):
        for j in range(kw):
            if kernel[i, j]:
                out = np.minimum(out, pad[i : i + h, j : j + w])
    return out


def dilate(src, kernel, iterations=1, **k):
    """Maximum over the kernel footprint anchored at its centre; cells outside the image do not
    take part (OpenCV's default border value for dilation is the type's minimum)."""
    assert iterations == 1
    kh, kw = kernel.shape
    ay, ax = kh // 2, kw // 2
    h, w = src.shape
    low = np.iinfo(src.dtype).min if np.issubdtype(src.dtype, np.integer) else -np.inf
    pad = np.full((h + kh, w + kw), low, dtype=src.dtype)
    pad[ay : ay + h, ax : ax + w] = src
    out = np.full((h, w), low, dtype=src.dtype)
    for i in range(kh):
        for j in range(kw):
            if kernel[i, j]:
                out = np.maximum(out, pad[i : i + h, j : j + w])
    return out


def _make_cv2():
    m = types.ModuleType("cv2")
    m.circle = circle
    m.erode = erode
    m.dilate = dilate
    m.getStructuringElement = lambda shape, ksize: np.ones((ksize[1], ksize[0]), dtype=np.uint8)
    m.line = line
    m.getRotationMatrix2D = lambda c, ang, s: np.zeros((2, 3))
    m.warpAffine = lambda src, M, dsize, **k: np.zeros_like(src)
    m.resize = lambda src, dsize, **k: np.zeros(
        (dsize[1], dsize[0]) + src.shape[2:], dtype=src.dtype
    )
    m.INTER_NEAREST, m.INTER_AREA, m.BORDER_CONSTANT, m.MORPH_RECT = 0, 3, 0, 0
    return m


# ----------------------------------------------------------------------------- habitat
def _make_habitat():
    hab = types.ModuleType("habitat")
    tasks = types.ModuleType("habitat.tasks")
    utils = types.ModuleType("habitat.tasks.utils")

    def cartesian_to_polar(x, y):
        rho = np.sqrt(x ** 2 + y ** 2)
        phi = np.arctan2(y, x)
        return rho, phi

    utils.cartesian_to_polar = cartesian_to_polar
    hab.tasks = tasks
    tasks.utils = utils
    return {"habitat": hab, "habitat.tasks": tasks, "habitat.tasks.utils": utils}


def install():
    sys.modules.setdefault("numpy_indexed", _make_numpy_indexed())
    sys.modules.setdefault("quaternion", _make_quaternion())
    sys.modules.setdefault("cv2", _make_cv2())
    for k, v in _make_habitat().items():
        sys.modules.setdefault(k, v)
