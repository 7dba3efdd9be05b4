"""Comparison helpers shared by the GPU parity tests.

The bar for floating-point state is 1e-5 relative (atol 1e-6 for values near zero).  Log-odds
cells are sums of log(p_c / p_0) over the frames: an entry can cancel to nearly zero while its
neighbours in the same cell are tens in magnitude, so their error is measured against the
cell's largest finite magnitude (`close_rows`)."""
import numpy as np

RTOL, ATOL = 1e-5, 1e-6
GOAL_IDS = (3, 4, 5, 6, 7, 8)


def _same_specials(got, want, name):
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"{name}: NaN pattern"
    inf = np.isinf(want)
    assert np.array_equal(np.isinf(got), inf), f"{name}: inf pattern"
    assert np.array_equal(got[inf], want[inf]), f"{name}: sign of an infinity"


def close(got, want, name, rtol=RTOL, atol=ATOL):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    _same_specials(got, want, name)
    fin = np.isfinite(want)
    err = np.abs(got[fin] - want[fin])
    tol = atol + rtol * np.abs(want[fin])
    assert (err <= tol).all(), f"{name}: max err {err.max():.3e} (ratio {(err / tol).max():.2f})"


def close_rows(got, want, name, rtol=RTOL, atol=ATOL):
    """|got - want| <= atol + rtol * (largest finite |want| of the same cell), last axis = classes."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    _same_specials(got, want, name)
    fin = np.isfinite(want)
    scale = np.max(np.where(fin, np.abs(want), 0.0), axis=-1, keepdims=True)
    err = np.where(fin, np.abs(np.where(fin, got, 0.0) - np.where(fin, want, 0.0)), 0.0)
    tol = atol + rtol * np.broadcast_to(scale, want.shape)
    assert (err <= tol).all(), f"{name}: max err {err.max():.3e} (ratio {(err / tol).max():.2f})"


def labels_agree(sem, want_sem, fused_want, name="semantic_map", margin=1e-5):
    """Labels must be equal, except where the reference's fused vector has its two leading values
    within `margin` relative of each other (the argmax is then decided by the last rounding);
    a goal label (ids 3..8) must never appear or disappear.  Returns the number of such cells."""
    bad = sem != want_sem
    if not bad.any():
        return 0
    top2 = np.sort(np.nan_to_num(fused_want[bad].astype(np.float64), nan=np.inf), axis=-1)[:, -2:]
    gap = np.abs(top2[:, 1] - top2[:, 0]) / np.maximum(np.abs(top2[:, 1]), 1e-30)
    assert (gap <= margin).all(), f"{name}: {bad.sum()} label mismatches, largest top-2 gap {gap.max():.2e}"
    goal = np.isin(sem[bad], GOAL_IDS) | np.isin(want_sem[bad], GOAL_IDS)
    assert not goal.any(), f"{name}: a goal label differs on {goal.sum()} cells"
    return int(bad.sum())


def elementwise_relative(got, want, floor=1e-3):
    """Element-wise relative error over the finite entries with |want| >= floor:
    (largest, fraction of those entries above 1e-5, number of entries)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    sel = np.isfinite(want) & np.isfinite(got) & (np.abs(want) >= floor)
    if not sel.any():
        return 0.0, 0.0, 0
    rel = np.abs(got[sel] - want[sel]) / np.abs(want[sel])
    return float(rel.max()), float((rel > 1e-5).mean()), int(sel.sum())
