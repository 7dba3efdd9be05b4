"""ctypes binding of include/ssmap.h.  Loading fails loudly when the HIP library is absent:
there is no CPU fallback for the product path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libssmap.so")

MAX_CLASSES = 155
MAX_POOL = 8
STATIC_SCRATCH_BYTES = 64 << 20

RULES = {"direct": 0, "argmax": 1, "goal_decay": 2, "avg_prob": 3, "w_avg_prob": 4,
         "w2_avg_prob": 5, "log_odds": 6, "hv": 7, "prob_hv": 8, "stubborn": 9}
ENV_INDEX_ERROR = 1

EXPORTS = ("ssmap_create", "ssmap_destroy", "ssmap_set_schema", "ssmap_set_palette",
           "ssmap_set_params", "ssmap_bind_env", "ssmap_step", "ssmap_step_noalloc", "ssmap_stage1",
           "ssmap_read_results", "ssmap_gt_goal_query", "ssmap_update_list", "ssmap_profile_begin", "ssmap_profile_read",
           "ssmap_set_egomap", "ssmap_ego_maps", "ssmap_set_nb_model", "ssmap_path_append", "ssmap_path_paint", "ssmap_last_error", "ssmap_version")


class Config(C.Structure):
    _fields_ = [("n_envs", C.c_int32), ("img_h", C.c_int32), ("img_w", C.c_int32),
                ("n_classes", C.c_int32), ("n_schema", C.c_int32), ("rule", C.c_int32),
                ("use_height_map", C.c_int32), ("max_map_h", C.c_int32), ("max_map_w", C.c_int32),
                ("depth_min", C.c_float), ("depth_max", C.c_float),
                ("fx", C.c_double), ("fy", C.c_double),
                ("pixel2meter", C.c_double), ("height_cutoff_m", C.c_double)]


class EnvMaps(C.Structure):
    _fields_ = [("map_h", C.c_int32), ("map_w", C.c_int32), ("lower", C.c_double * 3),
                ("semantic_map", C.c_void_p), ("semantic_map_color", C.c_void_p),
                ("height_map", C.c_void_p), ("gt_map", C.c_void_p), ("unc_map", C.c_void_p),
                ("prob_map", C.c_void_p), ("weight_map", C.c_void_p), ("goal_map", C.c_void_p),
                ("background_map", C.c_void_p), ("hits", C.c_void_p), ("views", C.c_void_p),
                ("total_view", C.c_void_p), ("cumulative_conf", C.c_void_p), ("max_conf", C.c_void_p),
                ("max_other_conf", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [("entropy_threshold", C.c_float), ("filter_goal_prob", C.c_float),
                ("use_filter", C.c_int32), ("goal_decay", C.c_float), ("goal_mark", C.c_float),
                ("hv_ratio", C.c_float), ("hv_min_views", C.c_float),
                ("hv_limit_obj_view_dist", C.c_double), ("found_dist_m", C.c_double),
                ("nb_stats", C.c_int32), ("pad", C.c_int32)]


class Frames(C.Structure):
    _fields_ = [("depth", C.c_void_p), ("logits", C.c_void_p), ("probs", C.c_void_p),
                ("temperature", C.c_float), ("semantic", C.c_void_p),
                ("semantic_unc", C.c_void_p), ("semantic_max_prob", C.c_void_p),
                ("semantic_gt", C.c_void_p), ("agent_pose", C.c_void_p),
                ("camera_pose", C.c_void_p), ("goal_sem_id", C.c_void_p), ("skip", C.c_void_p)]


_lib = None


MAX_AUX = 32


class EgomapConfig(C.Structure):
    _fields_ = [("small_crop_px", C.c_int32), ("large_crop_px", C.c_int32), ("offset_for_cut", C.c_int32),
                ("out_w", C.c_int32), ("out_h", C.c_int32), ("agent_radius_px", C.c_int32),
                ("aux_bin_px", C.c_int32), ("n_aux", C.c_int32), ("aux_points", C.c_int32 * (2 * MAX_AUX)),
                ("active_goal_color", C.c_uint8 * 3), ("arrow_color", C.c_uint8 * 3), ("pad", C.c_uint8 * 2)]


def load():
    """Returns the loaded library; raises if it was not built (python semantic-search_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    # SSMAP_LIB names another build of the same library (A/B timing of two builds on one box)
    path = os.environ.get("SSMAP_LIB") or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build the HIP engine first "
            "(python semantic-search_amd/build.py); there is no CPU fallback")
    lib = C.CDLL(path)
    vp, i32 = C.c_void_p, C.c_int32
    lib.ssmap_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.ssmap_destroy.argtypes = [vp]
    lib.ssmap_destroy.restype = None
    lib.ssmap_set_schema.argtypes = [vp, vp, i32, vp, vp]
    lib.ssmap_set_palette.argtypes = [vp, vp, i32]
    lib.ssmap_set_params.argtypes = [vp, C.POINTER(Params)]
    lib.ssmap_bind_env.argtypes = [vp, i32, C.POINTER(EnvMaps)]
    lib.ssmap_step.argtypes = [vp, C.POINTER(Frames), vp]
    lib.ssmap_step_noalloc.argtypes = [vp, C.POINTER(Frames), vp]
    lib.ssmap_stage1.argtypes = [vp, vp, C.c_float, vp, vp, vp, vp, vp]
    lib.ssmap_read_results.argtypes = [vp, vp, vp, vp, vp]
    lib.ssmap_gt_goal_query.argtypes = [vp, vp, vp, vp]
    lib.ssmap_update_list.argtypes = [vp, i32, C.POINTER(vp)]
    lib.ssmap_profile_begin.argtypes = [vp, i32]
    lib.ssmap_profile_read.argtypes = [vp, vp, vp, vp]
    lib.ssmap_set_egomap.argtypes = [vp, C.POINTER(EgomapConfig)]
    lib.ssmap_ego_maps.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ssmap_set_nb_model.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    lib.ssmap_path_append.argtypes = [vp, i32, vp, i32, vp]
    lib.ssmap_path_paint.argtypes = [vp, vp]
    lib.ssmap_last_error.restype = C.c_char_p
    lib.ssmap_version.restype = C.c_char_p
    for name in EXPORTS:
        if name not in ("ssmap_destroy", "ssmap_last_error", "ssmap_version"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


class EngineError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise EngineError(f"ssmap error {rc}: {load().ssmap_last_error().decode()}")
