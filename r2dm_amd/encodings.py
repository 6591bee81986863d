"""Constant coordinate encodings of the denoiser's input (reference: /root/reference/models/encoding.py:80-149,
/root/reference/models/efficient_unet.py:220-229,278-281).

The encoding depends on the ray angles only -- not on the step, not on the sample -- so it is evaluated ONCE on the
host when the weights are packed and enters the engine as the constant ``__cenc`` (coord_channels, H, W); its
convolution with ``in_conv`` is folded into a per-pixel bias map there (csrc/engine.hip).  Three encodings exist
upstream; all three are tables of closed-form functions of (phi, theta):

    fourier_features      [sin(f_k . (phi, theta) + p_k), cos(...)]      2 * (ceil(log2 H) + ceil(log2 W)) channels
    polar_coordinates     (phi, theta) themselves                         2 channels
    spherical_harmonics   the 25 real spherical harmonics of degree <= 4 of the unit ray direction
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Tuple

import torch
import torch.nn.functional as F

# Real spherical harmonics Y_l^m, l = 0..4, in the component order of encoding.py:40-77 (m = -l..l within a degree), as
# (normalisation constant, polynomial in the direction cosines).  The constants are sqrt((2l+1)/(4 pi) (l-|m|)!/(l+|m|)!)
# times the polynomial's own leading factor, in double precision; the polynomials are evaluated in float32 like upstream.
_SH: List[Tuple[float, Callable]] = [
    # l = 0
    (0.28209479177387814, lambda x, y, z: torch.ones_like(x)),
    # l = 1
    (0.4886025119029199, lambda x, y, z: y),
    (0.4886025119029199, lambda x, y, z: z),
    (0.4886025119029199, lambda x, y, z: x),
    # l = 2
    (1.0925484305920792, lambda x, y, z: x * y),
    (1.0925484305920792, lambda x, y, z: y * z),
    (None, lambda x, y, z: 0.9461746957575601 * (z * z) - 0.31539156525251999),
    (1.0925484305920792, lambda x, y, z: x * z),
    (0.5462742152960396, lambda x, y, z: x * x - y * y),
    # l = 3
    (0.5900435899266435, lambda x, y, z: y * (3 * (x * x) - y * y)),
    (2.890611442640554, lambda x, y, z: x * y * z),
    (0.4570457994644658, lambda x, y, z: y * (5 * (z * z) - 1)),
    (0.3731763325901154, lambda x, y, z: z * (5 * (z * z) - 3)),
    (0.4570457994644658, lambda x, y, z: x * (5 * (z * z) - 1)),
    (1.445305721320277, lambda x, y, z: z * (x * x - y * y)),
    (0.5900435899266435, lambda x, y, z: x * (x * x - 3 * (y * y))),
    # l = 4
    (2.5033429417967046, lambda x, y, z: x * y * (x * x - y * y)),
    (1.7701307697799304, lambda x, y, z: y * z * (3 * (x * x) - y * y)),
    (0.9461746957575601, lambda x, y, z: x * y * (7 * (z * z) - 1)),
    (0.6690465435572892, lambda x, y, z: y * z * (7 * (z * z) - 3)),
    (0.10578554691520431, lambda x, y, z: 35 * (z * z) * (z * z) - 30 * (z * z) + 3),
    (0.6690465435572892, lambda x, y, z: x * z * (7 * (z * z) - 3)),
    (0.47308734787878004, lambda x, y, z: (x * x - y * y) * (7 * (z * z) - 1)),
    (1.7701307697799304, lambda x, y, z: x * z * (x * x - 3 * (y * y))),
    (0.6258357354491761, lambda x, y, z: (x * x) * (x * x - 3 * (y * y)) - (y * y) * (3 * (x * x) - y * y)),
]
SH_LEVELS = 5  # efficient_unet.py:222


def coord_channels(encoding: Optional[str], resolution) -> int:
    if encoding is None:
        return 0
    if encoding == "fourier_features":
        H, W = resolution
        return 2 * (int(math.ceil(math.log2(H))) + int(math.ceil(math.log2(W))))
    if encoding == "polar_coordinates":
        return 2
    if encoding == "spherical_harmonics":
        return SH_LEVELS ** 2
    raise ValueError(f"unknown coords_encoding {encoding!r}")


@torch.no_grad()
def coords_constant(encoding: Optional[str], coords: torch.Tensor, freqs: Optional[torch.Tensor] = None,
                    phase: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """``coords`` (1, 2, H, W) = (phi, theta) in radians -> (coord_channels, H, W) float32 on the CPU.

    Host evaluation on purpose: Fourier arguments reach 2^9 pi and f*theta must be the exact float product before
    sin/cos; a GPU conv library does not guarantee that (measured: ~1e-5 on the U-Net output)."""
    if encoding is None:
        return None
    c = coords.detach().float().cpu()
    if encoding == "polar_coordinates":  # nn.Identity (efficient_unet.py:224-226)
        return c[0].clone()
    if encoding == "fourier_features":  # encoding.py:141-146
        z = F.conv2d(c, freqs.detach().float().cpu(), phase.detach().float().cpu())
        return torch.cat([z.sin(), z.cos()], dim=1)[0]
    if encoding == "spherical_harmonics":  # encoding.py:98-111: direction = (cos t cos p, -sin t cos p, sin p)
        phi, theta = c[0, 0], c[0, 1]
        x, y, z = torch.cos(theta) * torch.cos(phi), -torch.sin(theta) * torch.cos(phi), torch.sin(phi)
        out = [poly(x, y, z) if k is None else k * poly(x, y, z) for k, poly in _SH[: SH_LEVELS ** 2]]
        return torch.stack(out)
    raise ValueError(f"unknown coords_encoding {encoding!r}")
