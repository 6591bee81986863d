"""ctypes binding of libr2dm_hip.so (C ABI: include/r2dm_hip.h).

There is NO fallback: if the shared library is missing, or a tensor is not on a ROCm device,
every entry point raises.  PyTorch only owns memory and streams here.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_size_t, c_void_p
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("R2DM_HIP_LIB", os.path.join(_HERE, "libr2dm_hip.so"))


class R2DMError(RuntimeError):
    pass


class R2DMRangeError(R2DMError):
    """The fp16-operand paths' range guard tripped (r2dm_check_range): results of the forwards since the last check are not valid."""


class R2DMRangeFallback(R2DMRangeError):
    """The guard tripped at the end of a loop that cannot replay itself (a hand-written ``p_step`` loop under
    ``deferred_range_check``): the denoiser has switched to the wide-range operand split, the loop's results are not valid, and
    repeating the call gives them (``repaint`` does that itself, from the saved generator states)."""


class Config(Structure):
    _fields_ = [
        ("in_channels", c_int32),
        ("out_channels", c_int32),
        ("height", c_int32),
        ("width", c_int32),
        ("base_channels", c_int32),
        ("temb_channels", c_int32),
        ("channel_multiplier", c_int32 * 4),
        ("num_residual_blocks", c_int32 * 4),
        ("gn_num_groups", c_int32),
        ("gn_eps", c_float),
        ("attn_num_heads", c_int32),
        ("coord_channels", c_int32),
        ("max_batch", c_int32),
    ]


class TensorInfo(Structure):
    _fields_ = [("key", c_char_p), ("numel", c_int64)]


# name -> (restype, argtypes); exactly the symbols include/r2dm_hip.h declares
_P = c_void_p
SIGNATURES = {
    "r2dm_last_error": (c_char_p, []),
    "r2dm_version": (c_char_p, []),
    "r2dm_create": (c_int32, [POINTER(c_void_p), POINTER(Config)]),
    "r2dm_destroy": (None, [_P]),
    "r2dm_num_tensors": (c_int64, [_P]),
    "r2dm_tensor_at": (c_int32, [_P, c_int64, POINTER(TensorInfo)]),
    "r2dm_blob_bytes": (c_size_t, [_P]),
    "r2dm_blob_layout_hash": (ctypes.c_uint64, [_P]),
    "r2dm_bind_blob": (c_int32, [_P, _P, c_size_t]),
    "r2dm_load_tensor": (c_int32, [_P, c_int64, _P, c_int64, _P]),
    "r2dm_workspace_bytes": (c_size_t, [_P, c_int32]),
    "r2dm_unet_forward": (c_int32, [_P, _P, _P, _P, c_int32, _P, c_size_t, _P]),
    "r2dm_set_conv_pieces": (c_int32, [_P, c_int32]),
    "r2dm_check_range": (c_int32, [_P, _P]),
    "r2dm_test_raise_range_bound": (c_int32, [_P, c_float, _P]),
    "r2dm_range_sites": (c_int32, [_P, POINTER(c_float), c_int32, POINTER(c_int32)]),
    "r2dm_range_site_name": (c_char_p, [_P, c_int32]),
    "r2dm_profile_enable": (c_int32, [_P, c_int32]),
    "r2dm_profile_read": (c_int32, [_P, POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(c_int64)]),
    "r2dm_profile_read_classes": (c_int32, [_P, POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(c_int64)]),
    "r2dm_profile_event_overhead": (c_int32, [_P, _P, POINTER(ctypes.c_double), POINTER(ctypes.c_double)]),
    "r2dm_posterior_step": (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int64, c_int32, c_int32, c_float, _P]),
    "r2dm_repaint_blend": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int64, c_int32, c_int32, _P]),
    "r2dm_q_step": (c_int32, [_P, _P, _P, _P, c_int32, c_int64, _P]),
    "r2dm_lidar_postprocess": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, c_float, c_float, _P]),
    "r2dm_lidar_postprocess_fmt": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, c_float, c_float, c_int32, _P]),
    "r2dm_conv_packed_elems": (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "r2dm_conv2d_ring": (c_int32, [_P, _P, _P, _P, _P, c_int32, _P, _P, _P, c_int32, c_int32, c_int32, c_int32,
                                   c_int32, c_int32, _P]),
    "r2dm_group_norm_scratch_bytes": (c_size_t, [c_int32, c_int32]),
    "r2dm_group_norm_affine": (c_int32, [_P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32,
                                         c_float, _P]),
    "r2dm_affine_act": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int64, c_int32, _P]),
    "r2dm_fir_down2": (c_int32, [_P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    "r2dm_fir_down2_stat_slots": (c_int32, [c_int32, c_int32, c_int32, c_int32]),
    "r2dm_fir_down2_stats": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    "r2dm_fir_up2": (c_int32, [_P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    "r2dm_attention": (c_int32, [_P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    "r2dm_time_embedding": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, _P]),
}

_lib: Optional[ctypes.CDLL] = None


def lib() -> ctypes.CDLL:
    """Load (once) and type the shared library.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise R2DMError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or r2dm_amd/csrc/build.sh). r2dm_amd has no CPU / PyTorch fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise R2DMError(lib().r2dm_last_error().decode(errors="replace"))


def require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise R2DMError(
            f"{what} is on {t.device}: r2dm_amd runs the sampling path only as HIP kernels on an MI355X "
            "(device 'cuda' under PyTorch-ROCm) and has no CPU fallback")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def f32c(t: torch.Tensor) -> torch.Tensor:
    """fp32 + contiguous (no copy when already so)."""
    return t.detach().to(torch.float32).contiguous()


# ---- thin functional wrappers -----------------------------------------------------------------
def posterior_step(x_t, pred, noise, coef, mode: int, objective: int, clip: float) -> torch.Tensor:
    require_gpu(x_t, "x_t")
    x_t, pred, coef = f32c(x_t), f32c(pred), f32c(coef)
    noise = None if noise is None else f32c(noise)
    out = torch.empty_like(x_t)
    B = x_t.shape[0]
    with torch.cuda.device(x_t.device):
        check(lib().r2dm_posterior_step(ptr(x_t), ptr(pred), ptr(noise), ptr(coef), ptr(out), B,
                                        x_t.numel() // B, mode, objective, float(clip), stream_ptr(x_t.device)))
    return out


def repaint_blend(known, noise, unknown, mask, coef) -> torch.Tensor:
    """mask * (known*alpha + noise*sigma) + (1 - mask) * unknown; coef (B,2) = (alpha, sigma)."""
    require_gpu(known, "known")
    known, noise, unknown, coef = f32c(known), f32c(noise), f32c(unknown), f32c(coef)
    B, C = known.shape[0], known.shape[1]
    mask = f32c(mask.to(known.device).expand(B, -1, *known.shape[2:]))
    out = torch.empty_like(known)
    with torch.cuda.device(known.device):
        check(lib().r2dm_repaint_blend(ptr(known), ptr(noise), ptr(unknown), ptr(mask), ptr(coef), ptr(out), B,
                                       known.numel() // B, C, mask.shape[1], stream_ptr(known.device)))
    return out


def q_step(x_s, noise, coef) -> torch.Tensor:
    """x_s * a_ts + std * noise; coef (B,2) = (a_ts, std)."""
    require_gpu(x_s, "x_s")
    x_s, noise, coef = f32c(x_s), f32c(noise), f32c(coef)
    out = torch.empty_like(x_s)
    B = x_s.shape[0]
    with torch.cuda.device(x_s.device):
        check(lib().r2dm_q_step(ptr(x_s), ptr(noise), ptr(coef), ptr(out), B, x_s.numel() // B, stream_ptr(x_s.device)))
    return out


DEPTH_FORMATS = {"log_depth": 0, "inverse_depth": 1, "depth": 2}


def lidar_postprocess(x, ray_angles, min_depth: float, max_depth: float, depth_format: str = "log_depth") -> torch.Tensor:
    require_gpu(x, "x")
    if depth_format not in DEPTH_FORMATS:
        raise ValueError(f"unknown depth_format {depth_format!r}")
    x, ang = f32c(x), f32c(ray_angles).reshape(2, *x.shape[-2:])
    B, _, H, W = x.shape
    out = torch.empty(B, 5, H, W, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        check(lib().r2dm_lidar_postprocess_fmt(ptr(x), ptr(ang), ptr(out), B, H, W, float(min_depth), float(max_depth),
                                               DEPTH_FORMATS[depth_format], stream_ptr(x.device)))
    return out
