"""Deterministic synthetic checkpoints.

No pretrained R2DM weights exist offline, so parity tests and ``bench.py``
run on a synthetic checkpoint that has the reference's exact on-disk structure
(/root/reference/train.py:294-303: ``{cfg, weights, ema_weights, global_step, ...}``) and is a
pure function of ``(config, seed)``: every tensor is drawn from a numpy PCG64 stream seeded by
the CRC32 of its key, so the dev container and the GPU box regenerate identical weights and the
golden fixtures only need to store inputs and outputs.

The tensors the reference zero-initialises (conv2, attention out_proj, out_conv:
efficient_unet.py:39,84,267) are given non-zero values at half scale -- with them at zero a
fresh network outputs exactly 0 and every parity test would be vacuous.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Optional

import numpy as np
import torch

from .spec import Entry, UNetGeometry, unet_entries

INV_SQRT2 = 1.0 / math.sqrt(2.0)


def hdl64e_ray_angles(H: int, W: int) -> torch.Tensor:
    """Linear HDL-64E elevation/azimuth grid in radians, shape (1,2,H,W): what training writes
    into ``model.coords`` for spherical projections (/root/reference/train.py:100-101,
    /root/reference/utils/lidar.py:9-20)."""
    el = (1 - torch.arange(H) / H) * (3 - (-25)) + (-25)
    az = (1 - torch.arange(W) / W) * (180 - (-180)) + (-180)
    el, az = torch.meshgrid(el, az, indexing="ij")
    return torch.stack([el, az])[None].deg2rad()


def fourier_tables(H: int, W: int):
    """Frequency/phase buffers of the Fourier coordinate encoding
    (/root/reference/models/encoding.py:121-139)."""
    Lh, Lw = int(math.ceil(math.log2(H))), int(math.ceil(math.log2(W)))
    fh = torch.cat([torch.arange(Lh).exp2(), torch.zeros(Lw)])
    fw = torch.cat([torch.zeros(Lh), torch.arange(Lw).exp2()])
    return torch.stack([fh, fw], dim=-1)[..., None, None].float(), torch.zeros(Lh + Lw)


def _draw(e: Entry, g: UNetGeometry, seed: int) -> torch.Tensor:
    rs = np.random.Generator(np.random.PCG64([zlib.crc32(e.key.encode()), seed]))
    n = int(np.prod(e.shape)) if len(e.shape) else 1
    H, W = g.resolution

    def normal(std):
        return torch.from_numpy((rs.standard_normal(n) * std).astype(np.float32)).reshape(e.shape)

    r = e.role
    if r in ("conv_w", "conv_w_zero", "linear_w", "linear_w_zero"):
        fan_in = int(np.prod(e.shape[1:]))
        std = 1.0 / math.sqrt(fan_in)
        if r.endswith("_zero"):
            std *= 0.5
        return normal(std)
    if r == "bias":
        return normal(0.05)
    if r == "bias_zero":
        return normal(0.025)
    if r == "gn_w":
        return 1.0 + normal(0.1)
    if r == "gn_b":
        return normal(0.1)
    if r == "inv_sqrt2":
        return torch.tensor(INV_SQRT2).float()
    if r == "fir_down":
        return torch.tensor([1.0, 3.0, 3.0, 1.0]) / 8.0
    if r == "fir_up":
        return torch.tensor([1.0, 3.0, 3.0, 1.0]) / 4.0
    if r == "coords":
        return hdl64e_ray_angles(H, W)
    if r == "fourier_freqs":
        return fourier_tables(H, W)[0]
    if r == "fourier_phase":
        return fourier_tables(H, W)[1]
    raise KeyError(r)


def synthetic_state_dict(g: UNetGeometry, seed: int = 0, prefix: str = "model.") -> Dict[str, torch.Tensor]:
    sd: Dict[str, torch.Tensor] = {}
    if prefix:
        sd["_dummy"] = torch.tensor([])
    for e in unet_entries(g):
        sd[prefix + e.key] = _draw(e, g, seed)
    return sd


def default_cfg_dict(
    resolution=(64, 1024),
    base_channels: int = 64,
    prediction_type: str = "eps",
    timestep_type: str = "continuous",
    noise_schedule: str = "cosine",
    num_training_steps: Optional[int] = None,
    **model_overrides,
) -> dict:
    """``asdict(Config)`` of the reference's default configuration
    (/root/reference/utils/option.py:6-77) with the fields tests vary exposed; any key of the ``model`` section
    (``coords_encoding``, ``gn_num_groups``, ``attn_num_heads``, ``channel_multiplier`` ...) may be overridden by keyword."""
    cfg = {
        "data": {
            "dataset": "kitti_360",
            "depth_format": "log_depth",
            "projection": "spherical-1024",
            "train_depth": True,
            "train_reflectance": True,
            "resolution": tuple(resolution),
        },
        "model": {
            "architecture": "efficient_unet",
            "base_channels": base_channels,
            "temb_channels": None,
            "channel_multiplier": (1, 2, 4, 8),
            "num_residual_blocks": (3, 3, 3, 3),
            "gn_num_groups": 8,
            "gn_eps": 1e-6,
            "attn_num_heads": 8,
            "coords_encoding": "fourier_features",
            "dropout": 0.0,
        },
        "diffusion": {
            "num_training_steps": num_training_steps,
            "num_sampling_steps": 1024,
            "prediction_type": prediction_type,
            "loss_type": "l2",
            "noise_schedule": noise_schedule,
            "timestep_type": timestep_type,
        },
        "training": {},
    }
    unknown = set(model_overrides) - set(cfg["model"])
    if unknown:
        raise TypeError(f"unknown model options {sorted(unknown)}")
    cfg["model"].update(model_overrides)
    return cfg


def geometry_from_cfg(cfg: dict) -> UNetGeometry:
    m, d = cfg["model"], cfg["data"]
    return UNetGeometry.make(
        in_channels=int(bool(d["train_depth"])) + int(bool(d["train_reflectance"])),
        resolution=d["resolution"],
        base_channels=m["base_channels"],
        temb_channels=m["temb_channels"],
        channel_multiplier=m["channel_multiplier"],
        num_residual_blocks=m["num_residual_blocks"],
        gn_num_groups=m["gn_num_groups"],
        gn_eps=m["gn_eps"],
        attn_num_heads=m["attn_num_heads"],
        coords_encoding=m["coords_encoding"],
    )


def synthetic_checkpoint(seed: int = 0, **cfg_kwargs) -> dict:
    """A checkpoint dict ``setup_model`` accepts, identical in structure to the reference's."""
    cfg = default_cfg_dict(**cfg_kwargs)
    g = geometry_from_cfg(cfg)
    sd = synthetic_state_dict(g, seed)
    if cfg["diffusion"]["timestep_type"] == "discrete":
        from .diffusion import discrete_tables

        beta, ab, abp, snr = discrete_tables(cfg["diffusion"]["num_training_steps"],
                                             cfg["diffusion"]["noise_schedule"])
        v4 = lambda t: t[:, None, None, None]
        sd.update(beta=v4(beta), alpha_bar=v4(ab), alpha_bar_prev=v4(abp), snr=v4(snr))
    return {"cfg": cfg, "weights": sd, "ema_weights": sd, "global_step": 0}
