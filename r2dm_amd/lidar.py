"""LiDAR range-image conventions used right after sampling ("next" row (f).1 of SURVEY.md section 8).

API of /root/reference/utils/lidar.py:9-120 (``LiDARUtility``): normalisation, log/inverse/linear
depth coding, validity mask and range -> xyz projection along per-pixel ray angles.  The small
elementwise members stay torch expressions on whatever device the tensors live on (they are not on
the per-step hot path); ``postprocess`` -- the fused denormalize -> revert_depth -> to_xyz -> concat
of /root/reference/sample_and_save.py:52-57 -- is one HIP kernel (``r2dm_lidar_postprocess``).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .synthetic import hdl64e_ray_angles


def get_hdl64e_linear_ray_angles(H: int = 64, W: int = 2048, device="cpu"):
    return hdl64e_ray_angles(H, W).to(device)


class LiDARUtility(nn.Module):
    def __init__(self, resolution, depth_format: str, min_depth: float, max_depth: float, ray_angles=None):
        super().__init__()
        assert depth_format in ("log_depth", "inverse_depth", "depth")
        self.resolution = tuple(resolution)
        self.depth_format = depth_format
        self.min_depth = min_depth
        self.max_depth = max_depth
        if ray_angles is None:
            ray_angles = get_hdl64e_linear_ray_angles(*self.resolution)
        else:
            assert ray_angles.ndim == 4 and ray_angles.shape[1] == 2
        ray_angles = F.interpolate(ray_angles.float(), size=self.resolution, mode="nearest-exact")
        self.register_buffer("ray_angles", ray_angles.float())

    @staticmethod
    def denormalize(x):
        return (x + 1) / 2

    @staticmethod
    def normalize(x):
        return x * 2 - 1

    def get_mask(self, metric):
        return ((metric > self.min_depth) & (metric < self.max_depth)).float()

    @torch.no_grad()
    def to_xyz(self, metric):
        assert metric.dim() == 4
        mask = (metric > self.min_depth) & (metric < self.max_depth)
        phi, theta = self.ray_angles[:, [0]], self.ray_angles[:, [1]]
        xyz = torch.cat((metric * phi.cos() * theta.cos(), metric * phi.cos() * theta.sin(), metric * phi.sin()), dim=1)
        return xyz * mask.float()

    @torch.no_grad()
    def convert_depth(self, metric, mask=None, depth_format=None):
        depth_format = self.depth_format if depth_format is None else depth_format
        mask = self.get_mask(metric) if mask is None else mask
        if depth_format == "log_depth":
            normalized = torch.log2(metric + 1) / math.log2(self.max_depth + 1)
        elif depth_format == "inverse_depth":
            normalized = self.min_depth / metric.add(1e-8)
        elif depth_format == "depth":
            normalized = metric.div(self.max_depth)
        else:
            raise ValueError
        return normalized.clamp(0, 1) * mask

    @torch.no_grad()
    def revert_depth(self, normalized, image_format=None):
        image_format = self.depth_format if image_format is None else image_format
        if image_format == "log_depth":
            metric = torch.exp2(normalized * math.log2(self.max_depth + 1)) - 1
        elif image_format == "inverse_depth":
            metric = self.min_depth / normalized.add(1e-8)
        elif image_format == "depth":
            metric = normalized.mul(self.max_depth)
        else:
            raise ValueError
        return metric * self.get_mask(metric)

    @torch.no_grad()
    def postprocess(self, sample):
        """(B,2,H,W) in [-1,1] -> (B,5,H,W) [depth, x, y, z, reflectance]
        (= `postprocess` of /root/reference/sample_and_save.py:52-57), one HIP launch."""
        return _lib.lidar_postprocess(sample, self.ray_angles[0], self.min_depth, self.max_depth, self.depth_format)
