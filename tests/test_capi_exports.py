"""The C-ABI library loads on a machine without a GPU and exports every entry point that
include/ssmap.h declares (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "ssmap.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ssmap_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import semantic_search_amd as ssa
    names = declared_functions()
    assert "ssmap_step" in names and "ssmap_create" in names and len(names) >= 12
    lib = ctypes.CDLL(ssa._capi.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), f"{name} is declared in ssmap.h but not exported"
    assert set(ssa._capi.EXPORTS) == set(names)


def test_struct_sizes_match_the_header_layout():
    import semantic_search_amd as ssa
    c = ssa._capi
    assert ctypes.sizeof(c.Config) == 9 * 4 + 2 * 4 + 4 + 4 * 8      # 9 ints, 2 floats, pad, 4 doubles
    assert ctypes.sizeof(c.EnvMaps) == 8 + 3 * 8 + 15 * 8
    assert ctypes.sizeof(c.Params) == 7 * 4 + 4 + 2 * 8 + 2 * 4
    assert ctypes.sizeof(c.Frames) == 12 * 8


def test_engine_refuses_to_run_without_gpu():
    import pytest
    import torch
    import semantic_search_amd as ssa
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        ssa.MapEngine(1, (48, 64), depth_hfov=59.0, depth_range=(0.5, 5.0))


def test_create_rejects_configs_before_touching_the_gpu():
    """Argument checks of ssmap_create run before any HIP call, so they answer on a CPU-only box."""
    import semantic_search_amd as ssa
    c = ssa._capi
    lib = c.load()

    def cfg(**over):
        kw = dict(n_envs=1, img_h=48, img_w=64, n_classes=8, n_schema=9, rule=c.RULES["avg_prob"],
                  use_height_map=1, max_map_h=100, max_map_w=100, depth_min=0.5, depth_max=5.0,
                  fx=56.0, fy=56.0, pixel2meter=0.03, height_cutoff_m=0.1)
        kw.update(over)
        return c.Config(**kw)

    for bad, text in ((dict(img_w=0), "img_w >= 1"), (dict(fx=1e12), "fx, fy"), (dict(pixel2meter=0.0), "positive"),
                      (dict(rule=99), "prob_aggregate"), (dict(max_map_w=9000), "map dimensions")):
        handle = ctypes.c_void_p()
        rc = lib.ssmap_create(ctypes.byref(cfg(**bad)), ctypes.byref(handle))
        assert rc != 0 and not handle.value, bad
        assert text in lib.ssmap_last_error().decode(), (bad, lib.ssmap_last_error())
    assert ctypes.sizeof(c.EgomapConfig) == 8 * 4 + 2 * c.MAX_AUX * 4 + 8


def test_late_entry_points_reject_null_arguments_without_a_gpu():
    """ssmap_set_nb_model / ssmap_path_append / ssmap_path_paint check their arguments first."""
    import semantic_search_amd as ssa
    lib = ssa._capi.load()
    assert lib.ssmap_set_nb_model(None, 3, 2, None, None, None, None, None) == 1
    assert lib.ssmap_path_append(None, 0, None, 0, None) == 1
    assert lib.ssmap_path_paint(None, None) == 1
    assert "null" in lib.ssmap_last_error().decode()
