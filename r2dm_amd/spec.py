"""Flat parameter/buffer table of the Efficient U-Net denoiser.

The reference builds its network from nested nn.Module classes
(/root/reference/models/efficient_unet.py:23-267); what a drop-in must preserve is only the
resulting *state-dict layout* (268 keys).  Here that layout is a plain
table generated from the config -- the single source of truth for

  * the generic parameter tree ``r2dm_amd.unet.EfficientUNet`` registers,
  * the packed device blob handed to the HIP library (``r2dm_amd.packing``),
  * the synthetic checkpoints used by tests and ``bench.py``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Iterable, List, Optional, Tuple


@dataclass(frozen=True)
class Entry:
    key: str
    shape: Tuple[int, ...]
    kind: str  # "param" | "buffer"
    role: str  # how the default initialiser / synthetic generator treats it


def _n_tuple(x, n: int) -> tuple:
    if isinstance(x, Iterable):
        x = tuple(x)
        assert len(x) == n
        return x
    return (x,) * n


@dataclass(frozen=True)
class UNetGeometry:
    """Everything the spec, the packer and the C library need to know about one network."""

    in_channels: int
    out_channels: int
    resolution: Tuple[int, int]
    base_channels: int
    temb_channels: int
    channel_multiplier: Tuple[int, int, int, int]
    num_residual_blocks: Tuple[int, int, int, int]
    gn_num_groups: int
    gn_eps: float
    attn_num_heads: int
    coords_encoding: Optional[str]

    @staticmethod
    def make(
        in_channels: int,
        resolution,
        out_channels: Optional[int] = None,
        base_channels: int = 128,
        temb_channels: Optional[int] = None,
        channel_multiplier=(1, 2, 4, 8),
        num_residual_blocks=(3, 3, 3, 3),
        gn_num_groups: int = 8,
        gn_eps: float = 1e-6,
        attn_num_heads: int = 8,
        coords_encoding: Optional[str] = "fourier_features",
    ) -> "UNetGeometry":
        return UNetGeometry(
            in_channels=in_channels,
            out_channels=in_channels if out_channels is None else out_channels,
            resolution=_n_tuple(resolution, 2),
            base_channels=base_channels,
            temb_channels=base_channels * 4 if temb_channels is None else temb_channels,
            channel_multiplier=_n_tuple(channel_multiplier, 4),
            num_residual_blocks=_n_tuple(num_residual_blocks, 4),
            gn_num_groups=gn_num_groups,
            gn_eps=gn_eps,
            attn_num_heads=attn_num_heads,
            coords_encoding=coords_encoding,
        )

    # -- derived ------------------------------------------------------------------------
    @property
    def fourier_levels(self) -> Tuple[int, int]:
        H, W = self.resolution
        return int(math.ceil(math.log2(H))), int(math.ceil(math.log2(W)))

    @property
    def coord_channels(self) -> int:
        from .encodings import coord_channels

        return coord_channels(self.coords_encoding, self.resolution)

    @property
    def level_channels(self) -> List[int]:
        return [self.base_channels] + [self.base_channels * m for m in self.channel_multiplier]

    def blocks(self):
        """(name, in_ch, out_ch, n_res, down, up, attn) for the eight U-Net stages in execution
        order (efficient_unet.py:253-265)."""
        C, N = self.level_channels, self.num_residual_blocks
        return [
            ("d_block1", C[0], C[1], N[0], False, False, False),
            ("d_block2", C[1], C[2], N[1], True, False, False),
            ("d_block3", C[2], C[3], N[2], True, False, False),
            ("d_block4", C[3], C[4], N[3], True, False, True),
            ("u_block4", C[4], C[3], N[3], False, True, True),
            ("u_block3", 2 * C[3], C[2], N[2], False, True, False),
            ("u_block2", 2 * C[2], C[1], N[1], False, True, False),
            ("u_block1", 2 * C[1], C[0], N[0], False, False, False),
        ]


def unet_entries(g: UNetGeometry) -> List[Entry]:
    """State-dict entries of the denoiser in the reference's own enumeration order."""
    H, W = g.resolution
    T = g.temb_channels
    E: List[Entry] = [Entry("coords", (1, 2, H, W), "buffer", "coords")]
    if g.coords_encoding == "fourier_features":
        L = sum(g.fourier_levels)
        E += [
            Entry("coords_encoding.freqs", (L, 2, 1, 1), "buffer", "fourier_freqs"),
            Entry("coords_encoding.phase", (L,), "buffer", "fourier_phase"),
        ]
    E += [
        Entry("time_embedding.1.weight", (T, g.base_channels), "param", "linear_w"),
        Entry("time_embedding.1.bias", (T,), "param", "bias"),
        Entry("time_embedding.3.weight", (T, T), "param", "linear_w"),
        Entry("time_embedding.3.bias", (T,), "param", "bias"),
        Entry("in_conv.weight", (g.base_channels, g.in_channels + g.coord_channels, 3, 3), "param", "conv_w"),
        Entry("in_conv.bias", (g.base_channels,), "param", "bias"),
    ]
    for name, cin, cout, n_res, down, up, attn in g.blocks():
        if down:
            E += [
                Entry(f"{name}.downsample.0.weight", (cout, cin, 3, 3), "param", "conv_w"),
                Entry(f"{name}.downsample.0.bias", (cout,), "param", "bias"),
                Entry(f"{name}.downsample.1.kernel", (4,), "buffer", "fir_down"),
            ]
        for i in range(n_res):
            p = f"{name}.residual_blocks.{i}."
            ci = cout if (i != 0 or down) else cin
            E += [
                Entry(p + "scale", (), "buffer", "inv_sqrt2"),
                Entry(p + "norm1.weight", (ci,), "param", "gn_w"),
                Entry(p + "norm1.bias", (ci,), "param", "gn_b"),
                Entry(p + "conv1.weight", (cout, ci, 3, 3), "param", "conv_w"),
                Entry(p + "conv1.bias", (cout,), "param", "bias"),
                Entry(p + "norm2.proj.1.weight", (2 * cout, T), "param", "linear_w"),
                Entry(p + "norm2.proj.1.bias", (2 * cout,), "param", "bias"),
                Entry(p + "conv2.weight", (cout, cout, 3, 3), "param", "conv_w_zero"),
                Entry(p + "conv2.bias", (cout,), "param", "bias_zero"),
            ]
            if ci != cout:
                E += [
                    Entry(p + "skip.weight", (cout, ci, 1, 1), "param", "conv_w"),
                    Entry(p + "skip.bias", (cout,), "param", "bias"),
                ]
        if attn:
            p = f"{name}.self_attn_block."
            E += [
                Entry(p + "scale", (), "buffer", "inv_sqrt2"),
                Entry(p + "norm.weight", (cout,), "param", "gn_w"),
                Entry(p + "norm.bias", (cout,), "param", "gn_b"),
                Entry(p + "attn.in_proj_weight", (3 * cout, cout), "param", "linear_w"),
                Entry(p + "attn.in_proj_bias", (3 * cout,), "param", "bias"),
                Entry(p + "attn.out_proj.weight", (cout, cout), "param", "linear_w_zero"),
                Entry(p + "attn.out_proj.bias", (cout,), "param", "bias_zero"),
            ]
        if up:
            E += [
                Entry(f"{name}.upsample.0.kernel", (4,), "buffer", "fir_up"),
                Entry(f"{name}.upsample.1.weight", (cout, cout, 3, 3), "param", "conv_w"),
                Entry(f"{name}.upsample.1.bias", (cout,), "param", "bias"),
            ]
    E += [
        Entry("out_conv.weight", (g.out_channels, g.base_channels, 3, 3), "param", "conv_w_zero"),
        Entry("out_conv.bias", (g.out_channels,), "param", "bias_zero"),
    ]
    return E
