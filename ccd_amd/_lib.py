"""ctypes binding of libccd_hip.so (the C ABI of include/ccd_hip.h).

There is NO CPU fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
The library is built in-tree by `__graft_entry__.build()` / `ccd_amd/csrc/build.sh` as ccd_amd/libccd_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CCD_HIP_LIB") or os.path.join(_HERE, "libccd_hip.so")   # override: lab builds (tools/gemm_lab.py)

_handle = None            # ctypes.CDLL once loaded
_stream_override = None   # tests of the ABI may pin the stream argument

P = C.c_void_p
I = C.c_int
L = C.c_long
F = C.c_float
U64 = C.c_uint64

# name -> argtypes (restype is always int unless listed in _RESTYPES); mirrors include/ccd_hip.h
SIGNATURES = {
    "ccd_abi_version": [],
    "ccd_build_info": [],
    "ccd_policy_set": [C.c_char_p, I],
    "ccd_policy_get": [C.c_char_p, P],
    "ccd_gemm_nt": [P, L, P, L, I, I, I, I, P, L, P, L, P, P, L, P, I, P, L, F, I, P, I, P, P],
    "ccd_gemm_nt_resid_ln": [P, L, P, L, I, I, I, P, L, P, P, L, P, I, P, P, F, P, L, P, P, P],
    "ccd_gemm_tn": [P, L, P, L, I, I, I, I, P, L, F, I, P, I, P],
    "ccd_gemm_tn_colsum": [P, L, P, L, I, I, I, P, L, P, I, P],
    "ccd_gemm_tn_pair": [P, L, P, L, I, I, P, L, P, L, P, L, I, I, P, L, I, P],
    "ccd_gemm_tn_pair_ws": [P, L, P, L, I, I, P, L, P, L, P, L, I, I, P, L, I, P, L, P],
    "ccd_gemm_tn_pair_ws_floats": [I, I, I, I],
    "ccd_gemm_nt_lnbwd": [P, L, P, L, I, I, I, P, L, P, P, P, P, L, I, P, P, P, L, P, I, P, P],
    "ccd_gemm_nt_lnbwd_g16": [P, L, P, L, I, I, I, P, L, P, P, P, P, L, I, P, P, P, L, P, I, P, P],
    "ccd_gemm_nt_lnbwd_tap_g16": [P, L, P, L, I, I, I, P, L, P, P, P, P, L, I, P, P, P, L, P, I, P, P, L, P, P, P, P],
    "ccd_proj_mlp_fused": [P, L, P, L, P, P, L, P, P, P, P, L, P, L, P, P, P, L, P, P, L, P, P, I, P, L, P, P, F, P, L, P, P, P, L, P, P, P, L,
                           I, I, I, P],
    "ccd_proj_mlp_fused_gact": [P, L, P, L, P, P, L, P, P, P, P, L, P, L, P, P, P, L, P, P, L, P, P, I, P, L, P, P, F, P, L, P, P, P, L, P, L,
                                P, P, P, L, I, I, I, P],
    "ccd_mlp_bwd_fused": [P, L, P, L, P, L, P, L, P, L, P, P, L, P, P, P, P, L, I, P, P, P, L, P, I, P, I, I, I, P],
    "ccd_mlp_fused": [P, L, P, L, P, P, L, P, P, L, P, I, P, L, P, P, F, P, L, P, P, P, L, P, L, I, I, I, P],
    "ccd_ln_fwd": [P, P, P, P, P, P, I, I, F, P],
    "ccd_ln_bwd": [P, P, P, P, P, P, I, P, P, P, P, I, P, I, I, P],
    "ccd_ln_bwd_g16": [P, P, P, P, P, P, I, P, P, P, P, I, P, I, I, P],
    "ccd_attention_fwd": [P, P, P, I, I, F, P],
    "ccd_attention_bwd": [P, P, P, P, P, P, I, I, F, P, P, P, P, L, P],
    "ccd_attention_bwd_ws_floats": [I, I],
    "ccd_patch_embed_fwd": [P, P, P, P, P, I, I, P],
    "ccd_patch_embed_bwd": [P, P, P, P, P, P, P, I, I, P],
    "ccd_patch_embed_bwd_g16": [P, P, P, P, P, P, I, I, P],
    "ccd_small_matmul_f32": [P, P, P, I, I, I, I, I, P],
    "ccd_colsum_bf16": [P, L, I, I, P, I, P, P],
    "ccd_mirror_bf16": [P, I, I, P],
    "ccd_cast_bf16": [P, P, L, P],
    "ccd_scale_cast_rows": [P, P, P, I, L, I, P],
    "ccd_ccl_label": [P, P, I, P],
    "ccd_mask_to_idmap": [P, P, I, P],
    "ccd_seg_to_mask": [P, P, I, P],
    "ccd_kmeans2_mask": [P, P, P, P, I, P],
    "ccd_augment_views": [P, P, P, P, P, I, I, I, P, P, P, I, P, I, P],
    "ccd_warp_idmap": [P, P, I, P, I, P],
    "ccd_region_stats": [P, P, P, P, I, P],
    "ccd_select_scan": [P, I, P, P, P, P, P],
    "ccd_region_pool_fwd": [P, P, P, P, P, P, P, I, I, P],
    "ccd_region_pool_bwd": [P, P, P, P, P, P, P, I, I, P],
    "ccd_idmap_to_planes": [P, P, I, P],
    "ccd_planes_to_idmap": [P, P, I, P],
    "ccd_l2norm_fwd": [P, P, P, I, P, I, I, P],
    "ccd_l2norm_bwd": [P, P, P, P, I, P, I, I, P],
    "ccd_weightnorm_fwd": [P, P, P, P, P, I, I, P],
    "ccd_weightnorm_bwd": [P, P, P, P, P, P, I, I, P],
    "ccd_dino_loss_fwd": [P, P, P, I, P, I, F, F, P, P, P],
    "ccd_dino_loss_bwd": [P, P, P, I, P, I, F, F, P, F, P, P, P],
    "ccd_head_loss_ws_floats": [I, I],
    "ccd_head_loss_fwd": [P, L, P, L, P, L, P, L, P, I, I, P, I, F, F, P, P, P, P],
    "ccd_head_loss_bwd": [P, L, P, L, P, L, P, L, P, I, I, P, I, F, F, P, F, P, P, L, P],
    "ccd_colsum_f32": [P, I, P, I, I, P, P],
    "ccd_matvec_bf16": [P, L, P, I, I, P, P],
    "ccd_center_ema": [P, P, I, P, I, F, P],
    "ccd_seg_loss": [P, P, P, I, F, P, P, P],
    "ccd_seg_sumsq": [P, P, P, P, I, P, P],
    "ccd_adamw": [P, P, P, P, P, P, P, P, I, P, P, F, F, F, F, P],
    "ccd_clip_scale": [P, P, P, P, I, P, F, P],
    "ccd_ema": [P, P, P, L, F, F, P, P],
    "ccd_conv_gemm": [P, L, P, P, L, I, I, P, L, P, P, P, P],
    "ccd_conv_wgrad": [P, L, I, P, L, P, L, P, L, P],
    "ccd_im2col": [P, L, P, L, P, P],
    "ccd_bn_finalize": [P, F, F, F, P, P, P, I, P],
    "ccd_bn_relu_fwd": [P, L, P, P, P, P, L, L, I, P],
    "ccd_bn_relu_bwd_reduce": [P, L, P, L, P, P, P, P, L, I, P],
    "ccd_bn_relu_bwd_apply": [P, L, P, L, P, P, P, P, F, P, P, P, P, L, L, I, P],
    "ccd_cls_gather_fwd": [P, L, P, P, I, I, I, P],
    "ccd_cls_grad_cols": [P, P, I, I, I, P],
    "ccd_cls_tail_fwd": [P, L, P, P, P, P, P, P, I, I, I, I, P],
    "ccd_cls_tail_bwd_reduce": [P, P, L, P, P, P, P, P, P, I, I, I, I, P],
    "ccd_cls_tail_bwd_apply": [P, P, L, P, P, P, P, P, F, P, P, P, P, P, P, L, I, I, I, I, P],
    "ccd_permute4": [P, P, P, P, P, I, P],
    "ccd_permute4_multi": [P, I, I, P],
    "ccd_bn_finalize_multi": [P, I, P],
    "ccd_dropout": [P, I, P, P, I, L, U64, F, P],
    "ccd_droppath_scales": [P, P, I, I, U64, P, P],
    "ccd_dec_embed_fwd": [P, P, P, P, I, I, I, I, U64, F, P],
    "ccd_dec_embed_bwd": [P, P, P, I, I, I, I, U64, F, P],
    "ccd_dec_attn_fwd": [P, L, P, L, P, L, P, L, P, P, P, P, I, I, I, I, I, I, F, U64, F, P],
    "ccd_dec_attn_bwd": [P, L, P, L, P, L, P, P, L, P, P, P, I, I, I, I, I, I, F, U64, F, P, L, P, L, P, L, P],
    "ccd_tf_loss_fwd": [P, L, I, P, I, I, I, P, P, P],
    "ccd_tf_loss_bwd": [P, L, I, P, I, I, I, P, P, P, P, L, P],
    "ccd_greedy_step": [P, L, I, I, P, I, I, P, I, P],
}
_RESTYPES = {"ccd_build_info": C.c_char_p, "ccd_attention_bwd_ws_floats": L, "ccd_gemm_tn_pair_ws_floats": L, "ccd_head_loss_ws_floats": L}


def bind(lib: C.CDLL) -> C.CDLL:
    """Attach argtypes/restypes; raises AttributeError if the library lacks a declared symbol."""
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, I)
    return lib


def get() -> C.CDLL:
    global _handle
    if _handle is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                f"ccd_amd: HIP library not found at {LIB_PATH}. Build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). ccd_amd has no CPU fallback.")
        _handle = bind(C.CDLL(LIB_PATH))
    return _handle


def stream() -> int:
    if _stream_override is not None:
        return _stream_override
    return torch.cuda.current_stream().cuda_stream


def check(code: int, what: str):
    if code != 0:
        kind = {-1: "invalid argument", -2: "unsupported shape"}.get(code, f"hipError {code}")
        raise RuntimeError(f"ccd_amd: {what} failed: {kind}")


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()
