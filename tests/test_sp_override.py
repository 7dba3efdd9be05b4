"""The shortest-path policy's found override (env.py:564-605) against vectors recorded by executing
that block of the reference's env.py over reference-mapper maps
(tests/golden/make_sp_override_golden.py -> tests/golden/sp_override/*.npz): the CPU restatement,
the host logic of the package, and (gpu) the engine's two map questions through the C ABI."""
import glob
import json
import os

import numpy as np
import pytest

from tests import golden_util as gu

SP_DIR = os.path.join(gu.GOLDEN_DIR, "sp_override")
NAMES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(SP_DIR, "*.npz")))


def _load(name):
    d = np.load(os.path.join(SP_DIR, name + ".npz"))
    return d, json.loads(str(d["meta"])), gu.GoldenCase(name)


def _obs(case, meta, i):
    obs = case.obs(i)
    obs["objectgoal"] = np.array([meta["goal_index"]])
    return obs


def test_fixtures_cover_every_branch():
    seen = dict(fp=False, fn=False, found_near=False, true_but_far=False, nothing_mapped=False)
    for name in NAMES:
        d, meta, _ = _load(name)
        seen["fp"] |= bool(d["fp"].any())
        seen["fn"] |= bool(d["fn"].any())
        seen["found_near"] |= bool(d["is_found"].any())
        seen["true_but_far"] |= bool((d["true_found"] & ~d["is_found"]).any())
        seen["nothing_mapped"] |= bool((~d["anywhere"].astype(bool)).any())
    assert all(seen.values()), seen
    assert len(NAMES) >= 8


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_the_reference_block(name):
    from oracle import mapper_oracle as mo
    from oracle.sp_override_oracle import SpOverride
    d, meta, case = _load(name)
    mapper = mo.OracleMapper(schema_lut=case.meta["lut"], objnav_to_sem_pred=case.goal_classes(), **case.mapper_kwargs())
    for k, v in case.meta["attrs"].items():
        setattr(mapper, k, v)
    if meta["found_dist_m"] is not None:
        mapper.found_dist_m = meta["found_dist_m"]
    mapper.reset(np.array(case.meta["bounds"], dtype=np.float64))
    sp = SpOverride()
    for i in range(case.n_frames):
        obs = _obs(case, meta, i)
        true_found = bool(mapper.update(obs))
        assert true_found == bool(d["true_found"][i]), i
        goal_id = int(mapper.goal2sem_id[obs["objectgoal"]][0])
        in_disk, anywhere = sp.questions(mapper, obs["agent_position_and_rotation"][:3], goal_id)
        assert (in_disk, anywhere) == (bool(d["in_disk"][i]), bool(d["anywhere"][i])), i
        found = sp.apply(true_found, in_disk, anywhere, float(d["dist"][i]), meta["success"])
        assert (found, sp.fp_found, sp.fn_found) == (bool(d["is_found"][i]), bool(d["fp"][i]), bool(d["fn"][i])), i


@pytest.mark.parametrize("name", NAMES)
def test_host_logic_matches_the_reference_block(name):
    """sp_found_override (the array form MapEngine callers use) on the recorded questions."""
    import semantic_search_amd as ssa
    d, meta, _ = _load(name)
    fp, fn = False, False
    for i in range(len(d["dist"])):
        found, fp_new, fn_new = ssa.sp_found_override([d["true_found"][i]], [d["in_disk"][i]], [d["anywhere"][i]],
                                                      [d["dist"][i]], meta["success"], [fp])
        fp = bool(fp_new[0])
        if fn_new[0] is not None:
            fn = bool(fn_new[0])
        assert (bool(found[0]), fp, fn) == (bool(d["is_found"][i]), bool(d["fp"][i]), bool(d["fn"][i])), i


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_engine_matches_the_reference_block(name):
    import semantic_search_amd as ssa
    d, meta, case = _load(name)
    mapper = ssa.SemanticMapper(**case.mapper_kwargs())
    for k, v in case.meta["attrs"].items():
        setattr(mapper, k, v)
    if meta["found_dist_m"] is not None:
        mapper.found_dist_m = meta["found_dist_m"]
    if case.meta["lut"] is not None:
        lut = np.asarray(case.meta["lut"])
        mapper.map_seg_to_schema = lambda ids: lut[ids]
    if case.goal_classes() is not None:
        mapper.objnav_to_sem_pred = case.goal_classes()
    mapper.reset(np.array(case.meta["bounds"], dtype=np.float64))
    sp = ssa.SpPolicyFoundOverride()
    for i in range(case.n_frames):
        _, true_found = mapper.update_map(_obs(case, meta, i), None)
        assert bool(true_found) == bool(d["true_found"][i]), i
        in_disk, anywhere = mapper._engine.gt_goal_query()
        assert (bool(in_disk[0]), bool(anywhere[0])) == (bool(d["in_disk"][i]), bool(d["anywhere"][i])), i
        found = sp(mapper, true_found, float(d["dist"][i]), meta["success"])
        assert (found, sp.sp_fp_found, sp.sp_fn_found) == (bool(d["is_found"][i]), bool(d["fp"][i]), bool(d["fn"][i])), i


@pytest.mark.gpu
def test_query_refused_before_the_first_step():
    import semantic_search_amd as ssa
    from semantic_search_amd import _capi
    eng = ssa.MapEngine(1, (48, 64), num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0),
                        probablistic_mapping=False, prob_aggregate="height_map_argmax", max_map_hw=(150, 150))
    eng.reset_env(0, np.array([0.0, -0.2, 0.0, 4.5, 2.5, 4.5]))
    with pytest.raises(_capi.EngineError, match="before any ssmap_step"):
        eng.gt_goal_query()
    eng.close()
