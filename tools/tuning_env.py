"""Experiment scripts only (tools/): the kernel-selection switches as a tuning dict read from AIRFE_* environment variables, so that the round-1-4 command lines
(`AIRFE_OVERLAP_LINES=0 python tools/experiments/soak.py ...`) keep working.  libairfe.so itself reads no environment: the dict goes into airfe_cfg.tuning."""
import os

NAMES = {"AIRFE_FUSE_LG_BLOCK": "fuse_lg_block", "AIRFE_SMALL_MAX_M": "gemm_small_max_m", "AIRFE_GEMM8_MIN_M": "gemm8_min_m", "AIRFE_GEMMR_MIN_M": "gemmr_min_m",
         "AIRFE_GEMMR_WGS": "gemmr_wgs", "AIRFE_QKV_PAIR": "qkv_pair", "AIRFE_BLOCK_MIN_M": "block_min_m", "AIRFE_LGB_TOKENS": "lgb_tokens", "AIRFE_SG_KENC_GEMM": "sg_kenc_gemm",
         "AIRFE_FOLD_QKV": "fold_qkv", "AIRFE_OVERLAP_LINES": "overlap_lines", "AIRFE_KF_GRAPH": "kf_graph", "AIRFE_KF_SPEC_ROWS": "kf_spec_rows", "AIRFE_FUSE_DEC": "fuse_dec",
         "AIRFE_ASSIGN_FUSED": "assign_fused", "AIRFE_FOLD_OUT_PROJ": "fold_out_proj", "AIRFE_DESC_GATHER_STREAM": "desc_gather_stream"}


def tuning_from_env():
    t = {v: int(os.environ[k]) for k, v in NAMES.items() if k in os.environ}
    return t or None
