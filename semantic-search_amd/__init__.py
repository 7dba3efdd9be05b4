"""MI355X-native semantic-map fusion engine (drop-in for the fusion path of
Semantic-Search's `SemanticMapper`).

    from semantic_search_amd import SemanticMapper, MapEngine

`SemanticMapper` mirrors the reference class for one environment; `MapEngine` steps a
batch of environments held on one GPU.  Both call the HIP library built by
`semantic-search_amd/build.py` through the C ABI in include/ssmap.h.
"""
from semantic_search_amd import _capi  # noqa: F401
from semantic_search_amd.engine import MapEngine, RuleParams, colormap, focal_lengths, map_size  # noqa: F401
from semantic_search_amd.mapper import SemanticMapper, SpPolicyFoundOverride, sp_found_override  # noqa: F401
from semantic_search_amd.group import MapperGroup  # noqa: F401

__all__ = ["SemanticMapper", "MapperGroup", "SpPolicyFoundOverride", "sp_found_override", "MapEngine", "RuleParams", "colormap", "focal_lengths", "map_size"]
