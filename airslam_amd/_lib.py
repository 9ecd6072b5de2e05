"""ctypes binding of libairfe.so (include/airfe.h).  No fallback: if the HIP library is missing or no GPU is
visible the product path raises — there is no CPU implementation behind this package."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libairfe.so")


class Tuning(C.Structure):
    """airfe_tuning (include/airfe.h): kernel-selection overrides, -1 = the library's choice"""
    _fields_ = [(n, C.c_int) for n in (
        "fuse_lg_block", "gemm_small_max_m", "gemm8_min_m", "gemmr_min_m", "gemmr_wgs", "qkv_pair", "block_min_m", "lgb_tokens", "sg_kenc_gemm",
        "fold_qkv", "overlap_lines", "kf_graph", "kf_spec_rows", "fuse_dec", "assign_fused", "fold_out_proj", "desc_gather_stream", "copy_wgs")] + [("reserved", C.c_int * 5)]


class Cfg(C.Structure):
    _fields_ = [
        ("device", C.c_int), ("precision", C.c_int), ("max_batch", C.c_int), ("enc_chunk", C.c_int),
        ("max_keypoints", C.c_int), ("keypoint_threshold", C.c_float), ("remove_borders", C.c_int),
        ("nms_radius", C.c_int), ("line_threshold", C.c_float), ("line_length_threshold", C.c_float),
        ("matcher", C.c_int), ("image_width", C.c_int), ("image_height", C.c_int), ("sinkhorn_iters", C.c_int),
        ("superpoint_pack", C.c_char_p), ("plnet_s1_pack", C.c_char_p), ("lightglue_pack", C.c_char_p),
        ("superglue_pack", C.c_char_p), ("matcher_precision", C.c_int), ("line_precision", C.c_int), ("check_launches", C.c_int),
        ("tuning", C.POINTER(Tuning)),
    ]


class Stage0(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "juncs_pred", "lines_pred", "iskeep", "idx_junc_to_end_min", "idx_junc_to_end_max",
        "loi_features", "loi_features_thin", "loi_features_aux")]


class SeqPolicy(C.Structure):
    """airfe_seq_policy (include/airfe_seq.h)"""
    _fields_ = [("min_init_stereo_feature", C.c_int), ("min_num_match", C.c_int), ("max_num_match", C.c_int), ("tracking_point_rate", C.c_float),
                ("tracking_parallax_rate", C.c_float), ("min_x_diff", C.c_double), ("max_x_diff", C.c_double), ("max_y_diff", C.c_double),
                ("image_width", C.c_int), ("image_height", C.c_int)]


class SeqFrame(C.Structure):
    """airfe_seq_frame (include/airfe_seq.h)"""
    _fields_ = ([(n, C.c_int) for n in ("frame_type", "candidate", "promoted", "dropped", "enough_match", "good_stereo_point", "n_left", "n_right", "n_lines_left",
                                        "n_lines_right", "n_junctions", "n_stereo", "n_matches")] + [("_pad", C.c_int)]
                + [(n, C.c_void_p) for n in ("features_left", "features_right", "lines_left", "lines_right", "junctions", "stereo_idx", "stereo_score", "matches_idx",
                                             "matches_score")])


# name -> (restype, argtypes); every symbol include/*.h declares
SIGNATURES = {
    "airfe_copy_rows_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "airfe_pack_rows_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "airfe_seq_default_policy": (None, [C.POINTER(SeqPolicy)]),
    "airfe_seq_add_keyframe_check": (C.c_int, [C.POINTER(SeqPolicy), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "airfe_seq_good_stereo_points": (C.c_int, [C.POINTER(SeqPolicy), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "airfe_seq_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(SeqPolicy), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "airfe_seq_destroy": (None, [C.c_void_p]),
    "airfe_seq_last_error": (C.c_char_p, [C.c_void_p]),
    "airfe_seq_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t]),
    "airfe_seq_end": (C.c_int, [C.c_void_p, C.POINTER(SeqFrame)]),
    "airfe_seq_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.POINTER(SeqFrame)]),
    "airfe_seq_stream": (C.c_void_p, [C.c_void_p]),
    "airfe_seq_wall_split": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "airfe_default_cfg": (None, [C.POINTER(Cfg)]),
    "airfe_default_tuning": (None, [C.POINTER(Tuning)]),
    "airfe_debug_fail_next_launch": (C.c_int, [C.c_void_p, C.c_int]),
    "airfe_create": (C.c_int, [C.POINTER(Cfg), C.POINTER(C.c_void_p)]),
    "airfe_destroy": (None, [C.c_void_p]),
    "airfe_last_error": (C.c_char_p, [C.c_void_p]),
    "airfe_detect_points": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                      C.POINTER(C.c_int)]),
    "airfe_has_line_branch": (C.c_int, [C.c_void_p]),
    "airfe_detect_plnet": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(Stage0), C.c_void_p,
                                     C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p,
                                     C.c_int, C.POINTER(C.c_int), C.c_int]),
    "airfe_stereo_keyframe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                        C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int),
                                        C.POINTER(C.c_int), C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_int,
                                        C.POINTER(C.c_int)]),
    "airfe_stereo_keyframe_tracked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int),
                                                C.POINTER(C.c_int), C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_int,
                                                C.POINTER(C.c_int), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "airfe_track_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int),
                                    C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "airfe_promote_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_int,
                                      C.POINTER(C.c_int)]),
    "airfe_adopt_reference": (C.c_int, [C.c_void_p]),
    "airfe_match_lightglue": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_int, C.POINTER(C.c_int)]),
    "airfe_match_superglue": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "airfe_assign_points_to_lines": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_int, C.c_void_p]),
    "airfe_match_lines": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_void_p, C.c_int, C.c_void_p]),
    "airfe_bow_load": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "airfe_bow_transform": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "airfe_bow_transform_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "airfe_set_rectify_maps": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "airfe_rectify_detect_points": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                              C.POINTER(C.c_int)]),
    "airfe_rectify_batch_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p,
                                          C.c_int, C.c_size_t, C.c_void_p]),
    "airfe_detect_points_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t,
                                                C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "airfe_match_lightglue_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                  C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "airfe_match_superglue_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "airfe_stereo_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "airfe_detect_plnet_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p,
                                               C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                               C.c_int, C.c_void_p, C.c_void_p]),
    "airfe_stereo_plnet_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t,
                                               C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                               C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                               C.c_void_p]),
    "airfe_debug_plnet_j2l": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "airfe_assign_points_to_lines_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "airfe_match_lines_batch_dev": (C.c_int, [C.c_void_p] + [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "airfe_sync": (C.c_int, [C.c_void_p]),
    "airfe_superglue_status": (C.c_int, [C.c_void_p, C.c_void_p]),
    "airfe_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "airfe_profile_stages": (C.c_int, []),
    "airfe_profile_stage_name": (C.c_char_p, [C.c_int]),
    "airfe_profile_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "airfe_debug_detector_maps": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "airfe_debug_lightglue_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "airfe_debug_superglue_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "airfe_debug_lg_filter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "airfe_debug_sg_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "airfe_debug_plnet_stage0": (C.c_int, [C.c_void_p] + [C.c_void_p] * 10),
    "airfe_debug_plnet_s1": (C.c_int, [C.c_void_p, C.POINTER(Stage0), C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "airfe_debug_plnet_s1_last": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "airfe_debug_preprocess": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "airfe_debug_conv3x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_int, C.c_void_p]),
    "airfe_debug_trace": (C.c_int, [C.c_void_p, C.c_int]),
    "airfe_debug_trace_stop": (C.c_int, [C.c_void_p, C.c_int]),
    "airfe_debug_trace_buffer": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "airfe_debug_trace_slots": (C.c_int, [C.c_void_p]),
    "airfe_debug_trace_slot": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
    "airfe_debug_trace_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "airfe_debug_gemm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.c_void_p]),
    "airfe_debug_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m airslam_amd.build` "
                "(hipcc, gfx950).  airslam_amd has no CPU fallback.")
        # PyTorch-ROCm ships its own libamdhip64; with two HIP runtimes in one process the one that initialises second sees no
        # device.  Importing torch BEFORE the dlopen makes libairfe.so bind to the runtime torch uses (same soname), whatever order
        # the caller imports things in (__graft_entry__.build() loads the library before smoke() imports torch).
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)      # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib
