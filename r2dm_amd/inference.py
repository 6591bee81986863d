"""Model factory: checkpoint dict / file -> (ddpm, lidar_utils, cfg).

Same entry points, arguments and return values as /root/reference/utils/inference.py:16-114.
"""
from __future__ import annotations

from pathlib import Path

import torch

from .diffusion import ContinuousTimeGaussianDiffusion, DiscreteTimeGaussianDiffusion, GaussianDiffusion
from .lidar import LiDARUtility
from .option import Config
from .unet import EfficientUNet


def count_parameters(model: torch.nn.Module) -> int:
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def setup_model(ckpt, device="cpu", ema: bool = True, show_info: bool = True, compile: bool = False,
                max_batch: int = 8, precision: str = "fp32"):
    """Build the sampler from a checkpoint (utils/inference.py:20-110).

    ``max_batch`` (extension) tells the HIP engine which batch size to tile its layers for; ``precision`` (extension)
    is ``"fp32"`` (parity mode) or ``"bf16x2"`` (reduced precision, see ``EfficientUNet.set_precision``)."""
    if isinstance(ckpt, (str, Path)):
        ckpt = torch.load(ckpt, map_location="cpu")
    cfg = Config(**ckpt["cfg"])

    in_channels = int(bool(cfg.data.train_depth)) + int(bool(cfg.data.train_reflectance))
    if cfg.model.architecture == "efficient_unet":
        model = EfficientUNet(
            in_channels=in_channels,
            resolution=cfg.data.resolution,
            base_channels=cfg.model.base_channels,
            temb_channels=cfg.model.temb_channels,
            channel_multiplier=cfg.model.channel_multiplier,
            num_residual_blocks=cfg.model.num_residual_blocks,
            gn_num_groups=cfg.model.gn_num_groups,
            gn_eps=cfg.model.gn_eps,
            attn_num_heads=cfg.model.attn_num_heads,
            coords_encoding=cfg.model.coords_encoding,
            ring=True,
            max_batch=max_batch,
        )
    elif cfg.model.architecture == "refinenet":
        raise NotImplementedError("architecture='refinenet' (LiDARGen baseline, /root/reference/models/refinenet.py) "
                                  "is outside the built hot path; see DESIGN.md")
    else:
        raise ValueError(f"Unknown: {cfg.model.architecture}")

    if cfg.diffusion.timestep_type == "discrete":
        ddpm = DiscreteTimeGaussianDiffusion(
            model=model,
            loss_type=cfg.diffusion.loss_type,
            num_training_steps=cfg.diffusion.num_training_steps,
            prediction_type=cfg.diffusion.prediction_type,
            noise_schedule=cfg.diffusion.noise_schedule,
        )
    elif cfg.diffusion.timestep_type == "continuous":
        ddpm = ContinuousTimeGaussianDiffusion(
            model=model,
            loss_type=cfg.diffusion.loss_type,
            prediction_type=cfg.diffusion.prediction_type,
            noise_schedule=cfg.diffusion.noise_schedule,
        )
    else:
        raise ValueError(f"Unknown: {cfg.diffusion.timestep_type}")

    state_dict = ckpt["ema_weights"] if ema else ckpt["weights"]
    ddpm.load_state_dict(state_dict)
    ddpm.eval()
    ddpm.requires_grad_(False)
    ddpm.to(device)
    ddpm.model.set_precision(precision)

    if compile:
        ddpm.model = torch.compile(ddpm.model)  # the ctypes call is a graph break -> runs eagerly

    lidar_utils = LiDARUtility(
        resolution=cfg.data.resolution,
        depth_format=cfg.data.depth_format,
        min_depth=cfg.data.min_depth,
        max_depth=cfg.data.max_depth,
        ray_angles=model.coords,
    )
    lidar_utils.eval()
    lidar_utils.to(device)

    if show_info:
        print(
            *[
                f"resolution: {model.resolution}",
                f"model: {model.__class__.__name__}",
                f"ddpm: {ddpm.__class__.__name__}",
                f'#steps:  {ckpt["global_step"]:,}',
                f"#params: {sum(p.numel() for p in ddpm.parameters()):,}",
            ],
            sep="\n",
        )
    return ddpm, lidar_utils, cfg


def setup_rng(seeds, device):
    """One generator per sample, seeded by its global seed (utils/inference.py:113-114)."""
    return [torch.Generator(device=device).manual_seed(int(i)) for i in seeds]
