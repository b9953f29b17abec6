"""ctypes binding of ``libvg_hip.so`` (C ABI declared in ``include/vg_api.h``).

There is no CPU fallback: if the shared library is missing or a call fails, a RuntimeError is
raised.  All calls enqueue on torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvg_hip.so")

VG_F32, VG_BF16 = 0, 1
ACT_NONE, ACT_LRELU, ACT_SIGMOID = 0, 1, 2
STATS_REPLICAS = 32


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("n", "h", "w", "oh", "ow", "ci", "co", "xcs", "ycs", "r", "s", "stride", "pad", "dtype")]


_lib: Optional[C.CDLL] = None

_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float
_D = C.c_double
_DP = C.POINTER(ConvDesc)


class BnBwdStats(C.Structure):
    """vg_bn_bwd_stats of include/vg_api.h."""
    _fields_ = [(n, C.c_void_p) for n in ("x", "y", "mean", "invstd", "scale", "shift", "sums")] + [("cpad", C.c_int), ("act", C.c_int), ("mask_out", C.c_int)]


class AugItem(C.Structure):
    _fields_ = [("src", C.c_void_p)] + [(n, C.c_int) for n in
                                         ("height", "width", "flip_h", "flip_v", "crop_i", "crop_j", "crop_h", "crop_w")]


_AP = C.POINTER(AugItem)

# name -> (restype, argtypes); must list every symbol declared in include/vg_api.h
SIGNATURES = {
    "vg_last_error": (C.c_char_p, []),
    "vg_version": (_I, []),
    "vg_set_conv_kernel": (None, [_I]),
    "vg_set_wgrad_kernel": (None, [_I]),
    "vg_last_conv_kernel": (C.c_char_p, []),
    "vg_last_wgrad_kernel": (C.c_char_p, []),
    "vg_conv_packed_bytes": (C.c_size_t, [_DP, _I]),
    "vg_conv_pack_weight": (_I, [_DP, _I, _P, _P, _P]),
    "vg_conv_pack_item_bytes": (C.c_size_t, []),
    "vg_conv_pack_item": (_I, [_DP, _I, _P, _P, _P]),
    "vg_conv_pack_batch": (_I, [_P, _I, _I, _P]),
    "vg_conv_fwd": (_I, [_DP, _P, _P, _P, _P, _P, _I, _I, _P]),
    "vg_conv_splitk_ws_bytes": (C.c_size_t, [_DP]),
    "vg_conv_fwd_splitk": (_I, [_DP, _P, _P, _P, _P, _I, _I, _P, _P]),
    "vg_conv_dgrad": (_I, [_DP, _P, _P, _P, _P, _P, _I, _I, _P]),
    "vg_conv_dgrad_bnstats": (_I, [_DP, _P, _P, _P, _I, C.POINTER(BnBwdStats), C.POINTER(C.c_int), _P]),
    "vg_conv_fwd_bnstats": (_I, [_DP, _P, _P, _P, _I, C.POINTER(BnBwdStats), C.POINTER(C.c_int), _P]),
    "vg_conv_wgrad_ws_bytes": (C.c_size_t, [_DP]),
    "vg_conv_wgrad": (_I, [_DP, _P, _P, _P, _P, _I, _P]),
    "vg_thin_conv_supported": (_I, [_DP]),
    "vg_thin_conv_ws_bytes": (C.c_size_t, []),
    "vg_thin_conv_fwd": (_I, [_DP, _P, _P, _P, _P, _P, _I, _P, _P]),
    "vg_thin_conv_dgrad": (_I, [_DP, _P, _P, _P, _P, _P]),
    "vg_thin_conv_wgrad_ws_bytes": (C.c_size_t, [_DP]),
    "vg_thin_conv_wgrad": (_I, [_DP, _P, _P, _P, _P, _I, _P]),
    "vg_conv1_supported": (_I, [_DP]),
    "vg_conv1_fwd": (_I, [_DP, _P, _P, _P, _P, _I, _P]),
    "vg_conv1_wgrad_ws_bytes": (C.c_size_t, [_DP]),
    "vg_conv1_wgrad": (_I, [_DP, _P, _P, _P, _P, _I, _P]),
    "vg_linear_wgrad": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "vg_linear_dgrad": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "vg_set_deterministic": (None, [_I]),
    "vg_get_deterministic": (_I, []),
    "vg_channel_sum": (_I, [_P, _L, _I, _I, _I, _P, _P, _I, _P]),
    "vg_channel_sum_noalloc": (_I, [_P, _L, _I, _I, _I, _P, _I, _P]),
    "vg_nchw_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vg_nhwc_to_nchw": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vg_avgpool2_nchw": (_I, [_P, _P, _L, _I, _I, _P]),
    "vg_bn_finalize": (_I, [_P, _I, _I, _L, _P, _P, _F, _F, _P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "vg_channel_stats": (_I, [_P, _P, _L, _I, _I, _I, _I, _P]),
    "vg_stats_collapse": (_I, [_P, _I, _P, _P]),
    "vg_stats_load": (_I, [_P, _I, _P, _P]),
    "vg_bn_act_fwd": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P]),
    "vg_bn_bwd_reduce": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "vg_bn_bwd_reduce_dz": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "vg_bn_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _L, _L, _F, _I, _I, _I, _I, _I, _P]),
    "vg_act_bwd": (_I, [_P, _P, _P, _L, _I, _I, _P]),
    "vg_add_inplace": (_I, [_P, _P, _L, _I, _P]),
    "vg_vq_assign": (_I, [_P, _P, _L, _I, _I, _P, _P, _P]),
    "vg_vq_shortlist_ws_bytes": (C.c_size_t, [_L, _I, _I]),
    "vg_vq_assign_shortlist": (_I, [_P, _P, _L, _I, _I, _P, _P, _P, _P]),
    "vg_vq_gather": (_I, [_P, _P, _P, _L, _I, _I, _P, _P, _P, _P, _P]),
    "vg_vq_ema_update": (_I, [_P, _P, _P, _P, _P, _I, _I, _F, _P, _P]),
    "vg_vq_bwd": (_I, [_P, _P, _P, _P, _L, _F, _P, _P]),
    "vg_sqdiff_sum": (_I, [_P, _P, _L, _I, _I, _I, _P, _P]),
    "vg_sqdiff_bwd": (_I, [_P, _P, _P, _F, _L, _I, _I, _I, _P, _P]),
    "vg_reduce_f32": (_I, [_P, _L, _I, _P, _P]),
    "vg_reduce_f32_bwd": (_I, [_P, _P, _F, _L, _I, _P, _P]),
    "vg_reparam_kl_fwd": (_I, [_P, _P, _P, _L, _P, _P, _P]),
    "vg_reparam_kl_bwd": (_I, [_P, _P, _P, _P, _P, _F, _L, _P, _P, _P]),
    "vg_colsum_f32": (_I, [_P, _I, _L, _P, _P]),
    "vg_clamp": (_I, [_P, _P, _F, _F, _P, _L, _P]),
    "vg_scalar_finish": (_I, [_P, _D, _P, _P]),
    "vg_attn_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "vg_attn_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "vg_attn_supported_bf16": (_I, [_I, _I, _I]),
    "vg_attn_fwd_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "vg_attn_bwd_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "vg_gamma_residual": (_I, [_P, _P, _P, _P, _L, _P]),
    "vg_gamma_residual_bwd": (_I, [_P, _P, _P, _P, _P, _L, _P]),
    "vg_cast": (_I, [_P, _P, _L, _I, _I, _P]),
    "vg_adam_step": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _F, _P]),
    "vg_augment_ws_bytes": (C.c_size_t, [_AP, _I, _I, _I]),
    "vg_augment_batch": (_I, [_AP, _I, _I, _I, _P, C.c_size_t, _P, _P]),
    "vg_codebook_usage": (_I, [_P, _L, _I, _I, _P, _P, _P]),
    "vg_ac_table": (_I, [_P, _I, _P, _P]),
    "vg_ac_stream_capacity": (_L, [_L]),
    "vg_ac_encode": (_I, [_P, _I, _L, _P, _I, _P, _L, _P, _P, _P]),
    "vg_ac_decode": (_I, [_P, _P, _I, _L, _L, _P, _I, _P, _P, _P]),
}


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `make -C vae_gan_amd/csrc` "
                               "(there is no CPU fallback for the HIP path)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if os.environ.get("VG_CONV_KERNEL"):      # ablation: 1 = im2col-row, 2 = + spatial-tile, 3 = + ping-pong (default)
            lib.vg_set_conv_kernel(int(os.environ["VG_CONV_KERNEL"]))
        _lib = lib
    return _lib


def last_error() -> str:
    return load().vg_last_error().decode("utf-8", "replace")


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise RuntimeError(f"libvg_hip {what} failed (rc={rc}): {last_error()}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream() -> int:
    """hipStream_t of torch's current stream on the current device (every launch asks: the raw accessor costs
    ~0.5 us against ~9 us for torch.cuda.current_stream(), 1.7 ms of host time per VQ step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def vg_dtype(dt: torch.dtype) -> int:
    if dt == torch.bfloat16:
        return VG_BF16
    if dt == torch.float32:
        return VG_F32
    raise TypeError(f"unsupported activation dtype {dt}")


def kc_of(dt: torch.dtype) -> int:
    """Channel-stride granule: one 64-byte K chunk."""
    return 32 if dt == torch.bfloat16 else 16


def pad_channels(c: int, dt: torch.dtype) -> int:
    k = kc_of(dt)
    return (c + k - 1) // k * k
