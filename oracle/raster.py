"""TEST INFRASTRUCTURE ONLY -- CPU oracle helpers (rasterisation and morphology).

Restates the OpenCV 4.7 primitives the reference mapper calls on the hot path
(opencv-python==4.7.0.72 in the reference's environment.yaml; cv2 is absent from this
image, so parity at this seam is UNPINNED):

  * filled `cv2.circle` (mapper.py:737-743): integer midpoint circle of
    modules/imgproc/src/drawing.cpp `Circle()`, fill branch -- every scan line between
    the mirrored octant points is filled, clipped to the image.
  * `cv2.erode` with a rectangular kernel, default anchor (kernel centre = size // 2) and
    the default border (treated as +inf, i.e. ignored) (mapper.py:1134-1141).
  * the 7x7 "same" convolution + clamp used as a dilation (mapper.py:1142-1152).
"""
from __future__ import annotations

import numpy as np


def disk_halfwidths(radius: int) -> np.ndarray:
    """hw[k] = largest |dx| filled on the rows at vertical offset +-k from the centre."""
    radius = int(radius)
    hw = np.full(radius + 1, -1, dtype=np.int64)
    err, dx, dy = 0, radius, 0
    plus, minus = 1, 2 * radius - 1
    while dx >= dy:
        hw[dy] = max(hw[dy], dx)
        hw[dx] = max(hw[dx], dy)
        dy += 1
        err += plus
        plus += 2
        if err > 0:
            err -= minus
            dx -= 1
            minus -= 2
    return hw


def disk_mask(shape, center_row: int, center_col: int, radius: int) -> np.ndarray:
    """Boolean image equal to cv2.circle(zeros, (center_col, center_row), radius, 1, -1)."""
    h, w = shape
    out = np.zeros((h, w), dtype=bool)
    hw = disk_halfwidths(radius)
    for k in range(-int(radius), int(radius) + 1):
        r = center_row + k
        half = hw[abs(k)]
        if r < 0 or r >= h or half < 0:
            continue
        c0, c1 = max(center_col - half, 0), min(center_col + half, w - 1)
        if c0 <= c1:
            out[r, c0 : c1 + 1] = True
    return out


def erode_rect(mask: np.ndarray, size: int) -> np.ndarray:
    """Binary erosion by a size x size box anchored at (size//2, size//2); outside = set."""
    h, w = mask.shape
    a = size // 2
    pad = np.ones((h + size, w + size), dtype=bool)
    pad[a : a + h, a : a + w] = mask
    out = np.ones((h, w), dtype=bool)
    for i in range(size):
        for j in range(size):
            out &= pad[i : i + h, j : j + w]
    return out


def dilate_same(mask: np.ndarray, size: int) -> np.ndarray:
    """Binary dilation equal to clamp(conv2d(mask, ones(size,size), padding='same'), 0, 1)."""
    h, w = mask.shape
    total = size - 1
    lo = total // 2  # torch 'same': left = total // 2, right = total - left
    pad = np.zeros((h + total, w + total), dtype=bool)
    pad[lo : lo + h, lo : lo + w] = mask
    out = np.zeros((h, w), dtype=bool)
    for i in range(size):
        for j in range(size):
            out |= pad[i : i + h, j : j + w]
    return out


def dilate_max(img: np.ndarray, kh: int, kw: int) -> np.ndarray:
    """Grey dilation by a kh x kw box anchored at (kh//2, kw//2), OpenCV's `dilate` with the
    default border: cells outside the image do not take part in the maximum."""
    h, w = img.shape
    ay, ax = kh // 2, kw // 2
    low = np.iinfo(img.dtype).min if np.issubdtype(img.dtype, np.integer) else -np.inf
    pad = np.full((h + kh, w + kw), low, dtype=img.dtype)
    pad[ay : ay + h, ax : ax + w] = img
    out = np.full((h, w), low, dtype=img.dtype)
    for i in range(kh):
        for j in range(kw):
            out = np.maximum(out, pad[i : i + h, j : j + w])
    return out


def line_cells(p0, p1, thickness: int):
    """Cells (x, y) of a thick line from p0 to p1 (both (x, y) ints): the 8-connected centre line
    of OpenCV's LineIterator (drawing.cpp: left to right, the error term starting at dx - 2 dy),
    drawn with a square brush of `thickness` cells.  An APPROXIMATION of cv2.line for
    thickness > 1 (OpenCV fills a rotated rectangle and round caps in 16.16 fixed point, which
    is not restated here): the callers of the reference class use the real cv2.line when OpenCV
    is importable, this function otherwise and in the fixtures."""
    x0, y0, x1, y1 = int(p0[0]), int(p0[1]), int(p1[0]), int(p1[1])
    if x1 < x0:
        x0, y0, x1, y1 = x1, y1, x0, y0
    dx, dy = x1 - x0, y1 - y0
    ystep = -1 if dy < 0 else 1
    dy = abs(dy)
    pts = []
    if dy > dx:       # y is the major axis
        err, plus, minus = dy - 2 * dx, 2 * dy, -2 * dx
        x, y = x0, y0
        for _ in range(dy + 1):
            pts.append((x, y))
            m = err < 0
            err += minus + (plus if m else 0)
            y += ystep
            x += 1 if m else 0
    else:
        err, plus, minus = dx - 2 * dy, 2 * dx, -2 * dy
        x, y = x0, y0
        for _ in range(dx + 1):
            pts.append((x, y))
            m = err < 0
            err += minus + (plus if m else 0)
            x += 1
            y += ystep if m else 0
    lo, hi = -((thickness - 1) // 2), thickness // 2
    out = set()
    for x, y in pts:
        for oy in range(lo, hi + 1):
            for ox in range(lo, hi + 1):
                out.add((x + ox, y + oy))
    return sorted(out, key=lambda c: (c[1], c[0]))


def line(img: np.ndarray, p0, p1, color, thickness: int = 1):
    """cv2.line stand-in on a 2-D array (see line_cells); clipped to the image, returns img."""
    h, w = img.shape[:2]
    for x, y in line_cells(p0, p1, max(int(thickness), 1)):
        if 0 <= x < w and 0 <= y < h:
            img[y, x] = color
    return img
