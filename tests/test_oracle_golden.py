"""The CPU oracle against the golden vectors recorded from the reference's mapper.py.

Bit-exact on everything: the oracle uses the same numpy/scipy operations in the same
order as the reference for the float32 map arithmetic, and oracle/geometry.c for the
float64 geometry."""
import numpy as np
import pytest

from oracle import mapper_oracle as mo
from tests import golden_util as gu


@pytest.mark.parametrize("name", gu.case_names())
def test_oracle_matches_reference_golden(name):
    case = gu.GoldenCase(name)
    mapper = mo.OracleMapper(schema_lut=case.meta["lut"], objnav_to_sem_pred=case.goal_classes(),
                             **case.mapper_kwargs())
    for k, v in case.meta["attrs"].items():
        setattr(mapper, k, v)
    mapper.reset(np.array(case.meta["bounds"], dtype=np.float64))
    for i in range(case.n_frames):
        obs = case.obs(i)
        found = mapper.update(obs)
        assert np.array_equal(mapper.last_update_pixels, case.expect(i, "update_pixels")), f"frame {i}"
        assert (1 if found else 0) == int(case.expect(i, "found")), f"frame {i}"
        assert np.array_equal(mapper.semantic_map, case.expect(i, "semantic_map")), f"frame {i}"
        if i == 0:
            assert np.array_equal(obs["depth"], case.expect(0, "depth_after"))
        for nm in gu.FLOAT_STATE:
            val = getattr(mapper, nm, None)
            if val is not None:
                assert np.array_equal(gu.finite_sum(val), case.expect(i, "sum_" + nm)), (i, nm)
    for nm in gu.FINAL_NAMES:
        want = case.final(nm)
        if want is not None:
            assert np.array_equal(getattr(mapper, nm), want, equal_nan=True), nm
    for key in ("goal", "non_goal"):
        want = case.nb_rows(key)
        if want is not None:
            rows = mapper.eps_nb_data[key]
            assert [len(r) for r in rows] == list(want[1]), key
            got = np.concatenate(rows) if rows else np.zeros((0, 4))
            assert got.dtype == np.float64 and np.array_equal(got, want[0]), key


def test_naive_bayes_restatement_vetoes_some_goal_cells():
    """The stubborn fixtures really exercise both verdicts of the model."""
    case = gu.GoldenCase("stubborn_nb")
    model = case.mapper_kwargs()["nb_models"][3]
    tv, cc = case.final("total_view").reshape(-1), case.final("cumulative_conf").reshape(-1)
    seen = tv > 0
    feats = np.stack([tv[seen], cc[seen], case.final("max_conf").reshape(-1)[seen],
                      case.final("max_other_conf").reshape(-1)[seen]], axis=-1)
    pred = mo.gaussian_nb_predict(model, feats)
    assert (pred == 0).any() and (pred == 1).any()


def test_golden_cases_cover_edge_cases():
    """The fixture set really contains the quirks it is meant to pin."""
    seen_negative = seen_found = seen_sentinel = False
    for name in gu.case_names():
        case = gu.GoldenCase(name)
        for i in range(case.n_frames):
            seen_found |= bool(case.expect(i, "found"))
        d0 = case.data["f0_in_depth"]
        seen_sentinel |= bool(np.any((d0 == np.float32(0.5)) | (d0 == np.float32(5.0))))
        seen_negative |= name.startswith("negative_wrap")
    assert seen_negative and seen_found and seen_sentinel
