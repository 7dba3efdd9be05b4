"""HIP engine (through the C ABI, via the drop-in SemanticMapper) against the golden
vectors recorded from the reference's mapper.py.

Bar: update set (cell indices), semantic_map / gt_map / background_map / height_map and the
found flag bit-exact; float32 fused state within rtol 1e-5 (atol 1e-6 for values near 0).
"""
import numpy as np
import pytest

from tests import golden_util as gu

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-6
EXACT = ("semantic_map", "semantic_map_color", "gt_map", "background_map", "height_map", "total_view")


def _close(got, want, name):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    same_special = (np.isnan(got) == np.isnan(want)) & (np.isinf(got) == np.isinf(want))
    assert same_special.all(), f"{name}: nan/inf pattern differs at {np.argwhere(~same_special)[:5]}"
    fin = np.isfinite(want)
    assert np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)]), name
    err = np.abs(got[fin] - want[fin])
    tol = ATOL + RTOL * np.abs(want[fin])
    assert (err <= tol).all(), f"{name}: max err {err.max():.3e}, worst ratio {(err / tol).max():.2f}"


@pytest.mark.parametrize("name", gu.case_names())
def test_engine_matches_reference_golden(name):
    import semantic_search_amd as ssa

    case = gu.GoldenCase(name)
    mapper = ssa.SemanticMapper(**case.mapper_kwargs())
    for k, v in case.meta["attrs"].items():
        setattr(mapper, k, v)
    if case.meta["lut"] is not None:
        lut = np.asarray(case.meta["lut"])
        mapper.map_seg_to_schema = lambda ids: lut[ids]
    if case.goal_classes() is not None:
        mapper.objnav_to_sem_pred = case.goal_classes()
    mapper.reset(np.array(case.meta["bounds"], dtype=np.float64))
    for i in range(case.n_frames):
        obs = case.obs(i)
        _, found = mapper.update_map(obs, None)
        pix, _cells = mapper._engine.update_list(0, mapper.last_n_updates)
        assert np.array_equal(pix, case.expect(i, "update_pixels")), f"frame {i}: update set differs"
        assert (1 if found else 0) == int(case.expect(i, "found")), f"frame {i}: found flag"
        assert np.array_equal(mapper.semantic_map, case.expect(i, "semantic_map")), f"frame {i}: semantic_map"
        if i == 0:
            assert np.array_equal(obs["depth"], case.expect(0, "depth_after"))
    for nm in gu.FINAL_NAMES:
        want = case.final(nm)
        if want is None:
            continue
        got = getattr(mapper, nm)
        if nm in EXACT:
            assert np.array_equal(got, want), nm
        else:
            _close(got, want, nm)
    for key in ("goal", "non_goal"):
        want = case.nb_rows(key)
        if want is not None:
            rows = mapper.eps_nb_data[key]
            assert [len(r) for r in rows] == list(want[1]), key
            got = np.concatenate(rows) if rows else np.zeros((0, 4))
            assert got.dtype == np.float64
            _close(got, want[0], "eps_nb_data " + key)
