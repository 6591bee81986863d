"""CPU oracle for the R2DM sampling path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A clean-room, functional restatement (plain torch ops on whatever device/dtype the
inputs live on) of the reference's hot path:

    GaussianDiffusion.sample -> p_step -> EfficientUNet.forward

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product (``r2dm_amd``) never does and has no CPU fallback.

Pinning: ``tests/golden/make_golden.py`` imports the real reference (from
/root/reference, dev container only) and captures its inputs/outputs as golden vectors;
``tests/test_oracle_golden.py`` checks every function of this oracle against them.  The reference itself
ships no tests or golden vectors.

Every function names the reference file:line whose arithmetic it restates.  All
citations are relative to /root/reference.  Weights are read from a flat
``state_dict`` using the reference's own key names, with
the ``model.`` prefix of the diffusion wrapper stripped.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
INV_SQRT2 = 0.7071067811865476  # registered as float32 buffer `scale` (efficient_unet.py:29,66)


# --------------------------------------------------------------------------------------
# configuration (utils/option.py:6-29 defaults == the pretrained "config H")
# --------------------------------------------------------------------------------------
class UNetConfig:
    def __init__(
        self,
        resolution=(64, 1024),
        in_channels=2,
        base_channels=64,
        channel_multiplier=(1, 2, 4, 8),
        num_residual_blocks=(3, 3, 3, 3),
        gn_num_groups=8,
        gn_eps=1e-6,
        attn_num_heads=8,
        temb_channels=None,
    ):
        self.resolution = tuple(resolution)
        self.in_channels = in_channels
        self.base_channels = base_channels
        self.channel_multiplier = tuple(channel_multiplier)
        self.num_residual_blocks = tuple(num_residual_blocks)
        self.gn_num_groups = gn_num_groups
        self.gn_eps = gn_eps
        self.attn_num_heads = attn_num_heads
        self.temb_channels = base_channels * 4 if temb_channels is None else temb_channels


def strip_prefix(sd: Dict[str, Tensor], prefix: str = "model.") -> Dict[str, Tensor]:
    """ddpm.state_dict() keys are `model.<unet key>` plus `_dummy` (base.py:27,65)."""
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


# --------------------------------------------------------------------------------------
# L1 ops
# --------------------------------------------------------------------------------------
def ring_pad(x: Tensor, p: int) -> Tensor:
    """ops.py:39-43 -- circular padding along W (azimuth wraps), zeros along H."""
    if p == 0:
        return x
    x = torch.cat([x[..., -p:], x, x[..., :p]], dim=-1)
    return F.pad(x, (0, 0, p, p))


def conv_ring(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """ops.py:149-173 -- stride-1 cross-correlation; 3x3 gets ring padding 1, 1x1 none."""
    k = w.shape[-1]
    return F.conv2d(ring_pad(x, k // 2), w, b)


def fir_down2(x: Tensor) -> Tensor:
    """ops.py:52-143 with down=2: [1,3,3,1]/8 along W then H on the ring-padded (1) input,
    keeping every second sample."""
    k = torch.tensor([1.0, 3.0, 3.0, 1.0], dtype=x.dtype, device=x.device) / 8.0
    C = x.shape[1]
    xp = ring_pad(x, 1)
    xp = F.conv2d(xp, k.view(1, 1, 1, 4).repeat(C, 1, 1, 1), groups=C)
    xp = F.conv2d(xp, k.view(1, 1, 4, 1).repeat(C, 1, 1, 1), groups=C)
    return xp[:, :, ::2, ::2]


def fir_up2(x: Tensor) -> Tensor:
    """ops.py:52-143 with up=2: zero-insertion then [1,3,3,1]/4 per axis, which is, per axis,
    y[2i] = x[i-1]/4 + 3x[i]/4 ; y[2i+1] = 3x[i]/4 + x[i+1]/4 with wrap in W and zeros
    outside in H.  W pass first, then H (ops.py:131-132)."""
    xw = ring_pad(x, 1)  # (B,C,H+2,W+2)
    a, c, b = xw[..., :-2], xw[..., 1:-1], xw[..., 2:]
    even = a * 0.25 + c * 0.75
    odd = c * 0.75 + b * 0.25
    xw = torch.stack([even, odd], dim=-1).flatten(-2)  # (B,C,H+2,2W)
    a, c, b = xw[..., :-2, :], xw[..., 1:-1, :], xw[..., 2:, :]
    even = a * 0.25 + c * 0.75
    odd = c * 0.75 + b * 0.25
    return torch.stack([even, odd], dim=-2).flatten(-3, -2)  # (B,C,2H,2W)


def silu(x: Tensor) -> Tensor:
    return x * torch.sigmoid(x)


def group_norm(x: Tensor, groups: int, eps: float, w: Optional[Tensor], b: Optional[Tensor]) -> Tensor:
    """nn.GroupNorm semantics (efficient_unet.py:33,72): biased variance over (C/G,H,W)."""
    B, C = x.shape[:2]
    xg = x.reshape(B, groups, -1)
    mean = xg.mean(dim=-1, keepdim=True)
    var = xg.var(dim=-1, unbiased=False, keepdim=True)
    y = ((xg - mean) * torch.rsqrt(var + eps)).reshape(x.shape)
    if w is not None:
        y = y * w.view(1, C, 1, 1) + b.view(1, C, 1, 1)
    return y


def sinusoidal_embedding(t: Tensor, channels: int, max_period: float = 10_000.0) -> Tensor:
    """ops.py:14-29 -- [sin(t f_k), cos(t f_k)], f_k = exp(-ln(max_period) k/(channels/2-1))."""
    half = channels // 2
    f = torch.exp(-math.log(max_period) / (half - 1) * torch.arange(half, device=t.device))
    arg = t[:, None] * f[None, :]
    return torch.cat([arg.sin(), arg.cos()], dim=-1).to(t)


def fourier_features(coords: Tensor, freqs: Tensor, phase: Tensor) -> Tensor:
    """encoding.py:141-146 -- 1x1 conv of (phi, theta) by the frequency table, then [sin, cos]."""
    z = F.conv2d(coords, freqs.to(coords), phase.to(coords))
    return torch.cat([z.sin(), z.cos()], dim=1)


def polar_coords(H: int, W: int) -> Tensor:
    """encoding.py:80-89 -- default `coords` buffer (checkpoints overwrite it, train.py:100-109)."""
    phi = (0.5 - torch.arange(H) / H) * math.pi
    theta = (1 - torch.arange(W) / W) * 2 * math.pi - math.pi
    phi, theta = torch.meshgrid(phi, theta, indexing="ij")
    return torch.stack([phi, theta])[None]


def fourier_tables(H: int, W: int):
    """encoding.py:121-139 -- powers of two up to the resolution along each axis."""
    Lh, Lw = int(math.ceil(math.log2(H))), int(math.ceil(math.log2(W)))
    fh = torch.cat([torch.arange(Lh).exp2(), torch.zeros(Lw)])
    fw = torch.cat([torch.zeros(Lh), torch.arange(Lw).exp2()])
    freqs = torch.stack([fh, fw], dim=-1)[..., None, None]
    return freqs, torch.zeros(Lh + Lw)


# --------------------------------------------------------------------------------------
# L2 network
# --------------------------------------------------------------------------------------
def time_embedding(sd, cfg: UNetConfig, cond: Tensor) -> Tensor:
    """efficient_unet.py:232-237,275 -- sinusoidal(base) -> Linear -> SiLU -> Linear."""
    h = sinusoidal_embedding(cond, cfg.base_channels)
    h = F.linear(h, sd["time_embedding.1.weight"], sd["time_embedding.1.bias"])
    return F.linear(silu(h), sd["time_embedding.3.weight"], sd["time_embedding.3.bias"])


def residual_block(sd, p: str, cfg: UNetConfig, x: Tensor, temb: Tensor) -> Tensor:
    """efficient_unet.py:95-110 (+ AdaGN ops.py:196-200)."""
    G, eps = cfg.gn_num_groups, cfg.gn_eps
    h = silu(group_norm(x, G, eps, sd[p + "norm1.weight"], sd[p + "norm1.bias"]))
    h = conv_ring(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"])
    ss = F.linear(silu(temb), sd[p + "norm2.proj.1.weight"], sd[p + "norm2.proj.1.bias"])
    scale, shift = ss[:, :, None, None].chunk(2, dim=1)
    h = group_norm(h, G, eps, None, None) * (1 + scale) + shift
    h = conv_ring(silu(h), sd[p + "conv2.weight"], sd[p + "conv2.bias"])
    if (p + "skip.weight") in sd:
        x = conv_ring(x, sd[p + "skip.weight"], sd[p + "skip.bias"])
    return (x + h) * sd[p + "scale"]


def self_attention_block(sd, p: str, cfg: UNetConfig, x: Tensor) -> Tensor:
    """efficient_unet.py:42-53 with nn.MultiheadAttention spelled out."""
    B, C, H, W = x.shape
    nh = cfg.attn_num_heads
    d = C // nh
    h = group_norm(x, cfg.gn_num_groups, cfg.gn_eps, sd[p + "norm.weight"], sd[p + "norm.bias"])
    tok = h.flatten(2).transpose(1, 2)  # (B,N,C)
    qkv = F.linear(tok, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    split = lambda t: t.reshape(B, -1, nh, d).transpose(1, 2)  # (B,nh,N,d)
    q, k, v = split(q), split(k), split(v)
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(d), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, -1, C)
    o = F.linear(o, sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"])
    o = o.transpose(1, 2).reshape(B, C, H, W)
    return (x + o) * sd[p + "scale"]


def block(sd, p: str, cfg: UNetConfig, n_res: int, h: Tensor, temb: Tensor) -> Tensor:
    """efficient_unet.py:178-185 -- [conv->down] -> residual blocks -> [attn] -> [up->conv]."""
    if (p + "downsample.0.weight") in sd:
        h = conv_ring(h, sd[p + "downsample.0.weight"], sd[p + "downsample.0.bias"])
        h = fir_down2(h)
    for i in range(n_res):
        h = residual_block(sd, f"{p}residual_blocks.{i}.", cfg, h, temb)
    if (p + "self_attn_block.scale") in sd:
        h = self_attention_block(sd, p + "self_attn_block.", cfg, h)
    if (p + "upsample.1.weight") in sd:
        h = fir_up2(h)
        h = conv_ring(h, sd[p + "upsample.1.weight"], sd[p + "upsample.1.bias"])
    return h


def unet_forward(sd: Dict[str, Tensor], cfg: UNetConfig, x: Tensor, cond: Tensor) -> Tensor:
    """efficient_unet.py:269-295.  `sd` uses un-prefixed U-Net keys (see strip_prefix)."""
    B = x.shape[0]
    if cond.ndim == 0:
        cond = cond[None].repeat_interleave(B, dim=0)
    temb = time_embedding(sd, cfg, cond.to(x))
    cenc = fourier_features(sd["coords"].to(x), sd["coords_encoding.freqs"], sd["coords_encoding.phase"])
    h = torch.cat([x, cenc.repeat_interleave(B, dim=0)], dim=1)
    h = conv_ring(h, sd["in_conv.weight"], sd["in_conv.bias"])
    N = cfg.num_residual_blocks
    h1 = block(sd, "d_block1.", cfg, N[0], h, temb)
    h2 = block(sd, "d_block2.", cfg, N[1], h1, temb)
    h3 = block(sd, "d_block3.", cfg, N[2], h2, temb)
    h4 = block(sd, "d_block4.", cfg, N[3], h3, temb)
    h = block(sd, "u_block4.", cfg, N[3], h4, temb)
    h = block(sd, "u_block3.", cfg, N[2], torch.cat([h, h3], 1), temb)
    h = block(sd, "u_block2.", cfg, N[1], torch.cat([h, h2], 1), temb)
    h = block(sd, "u_block1.", cfg, N[0], torch.cat([h, h1], 1), temb)
    return conv_ring(h, sd["out_conv.weight"], sd["out_conv.bias"])


# --------------------------------------------------------------------------------------
# L3 diffusion (continuous time)
# --------------------------------------------------------------------------------------
def log_snr_cosine(t: Tensor, logsnr_min: float = -15.0, logsnr_max: float = 15.0) -> Tensor:
    """continuous_time.py:14-15,22-29."""
    t_min = math.atan(math.exp(-0.5 * logsnr_max))
    t_max = math.atan(math.exp(-0.5 * logsnr_min))
    return -2 * torch.log(torch.tan(t_min + t * (t_max - t_min)).clamp(min=1e-20))


def log_snr_linear(t: Tensor) -> Tensor:
    """continuous_time.py:18-19."""
    return -torch.log(torch.special.expm1(1e-4 + 10 * (t**2)).clamp(min=1e-20))


def alpha_sigma(log_snr: Tensor):
    """continuous_time.py:61-63."""
    return log_snr.sigmoid().sqrt(), (-log_snr).sigmoid().sqrt()


def draw_noise(shape: Sequence[int], rng, device, dtype=torch.float32) -> Tensor:
    """base.py:71-94 -- None / one Generator / one Generator per sample."""
    if rng is None:
        return torch.randn(*shape, device=device, dtype=dtype)
    if isinstance(rng, torch.Generator):
        return torch.randn(*shape, generator=rng, device=device, dtype=dtype)
    assert len(rng) == shape[0]
    return torch.stack([torch.randn(*shape[1:], generator=r, device=device, dtype=dtype) for r in rng])


def p_step_continuous(
    denoise,
    x_t: Tensor,
    step_t: Tensor,
    step_s: Tensor,
    noise: Tensor,
    mode: str = "ddpm",
    ddim_eta: float = 0.0,
    objective: str = "eps",
    log_snr=log_snr_cosine,
    clip: Optional[float] = 1.0,
) -> Tensor:
    """continuous_time.py:192-232 with the noise tensor passed in (teacher forcing).
    `denoise(x, cond)` is the network; cond is log-SNR(t) (continuous_time.py:207)."""
    v4 = lambda t: t[:, None, None, None]
    lt, ls = v4(log_snr(step_t)), v4(log_snr(step_s))
    a_t, s_t = alpha_sigma(lt)
    a_s, s_s = alpha_sigma(ls)
    pred = denoise(x_t, lt[:, 0, 0, 0])
    if objective == "eps":
        x_0 = (x_t - s_t * pred) / a_t
    elif objective == "v":
        x_0 = a_t * x_t - s_t * pred
    elif objective == "x_0":
        x_0 = pred
    else:
        raise ValueError(objective)
    if clip is not None:
        x_0 = x_0.clamp(-clip, clip)
    if mode == "ddpm":
        c = -torch.special.expm1(lt - ls)
        mean = a_s * (x_t * (1 - c) / a_t + c * x_0)
        return mean + s_s * c.sqrt() * noise
    if mode == "ddim":
        c1 = ddim_eta * s_s / s_t * (1 - a_t**2 / a_s**2).sqrt()
        c2 = (1 - a_s**2 - c1**2).sqrt()
        eps = (x_t - a_t * x_0) / s_t
        return a_s * x_0 + c1 * noise + c2 * eps
    raise ValueError(mode)


def sample_continuous(
    denoise,
    shape: Sequence[int],
    num_steps: int,
    rng=None,
    noises: Optional[List[Tensor]] = None,
    return_all: bool = False,
    mode: str = "ddpm",
    ddim_eta: float = 0.0,
    objective: str = "eps",
    device="cpu",
    dtype=torch.float32,
    log_snr=log_snr_cosine,
):
    """continuous_time.py:234-258.  Draw order: one initial draw + one per step (also for
    DDIM and on the last step).  `noises` (S+1 tensors) overrides the RNG for teacher forcing."""
    B = shape[0]
    nxt = (lambda i: noises[i].to(device=device, dtype=dtype)) if noises is not None else (
        lambda i: draw_noise(shape, rng, device, dtype))
    x = nxt(0)
    out = [x]
    steps = torch.linspace(1.0, 0.0, num_steps + 1, device=device)[None].repeat_interleave(B, dim=0)
    for i in range(num_steps):
        x = p_step_continuous(denoise, x, steps[:, i], steps[:, i + 1], nxt(i + 1), mode, ddim_eta,
                              objective, log_snr)
        out.append(x)
    return torch.stack(out) if return_all else x


def q_step_from_x_0(x_0: Tensor, step_t: Tensor, noise: Tensor, log_snr=log_snr_cosine) -> Tensor:
    """continuous_time.py:169-176: forward process q(z_t | x_0) with the noise passed in."""
    a, s = alpha_sigma(log_snr(step_t)[:, None, None, None])
    return x_0 * a + noise * s


def q_step(x_s: Tensor, step_t: Tensor, step_s: Tensor, noise: Tensor, log_snr=log_snr_cosine) -> Tensor:
    """continuous_time.py:178-190: q(z_t | z_s), 0 < s < t < 1."""
    v4 = lambda t: t[:, None, None, None]
    a_t, s_t = alpha_sigma(v4(log_snr(step_t)))
    a_s, s_s = alpha_sigma(v4(log_snr(step_s)))
    a_ts = a_t / a_s
    var = s_t.pow(2) - a_ts.pow(2) * s_s.pow(2)
    return x_s * a_ts + var.sqrt() * noise


def repaint_continuous(denoise, known: Tensor, mask: Tensor, num_steps: int, noises: List[Tensor], num_resample_steps: int = 1,
                       jump_length: int = 1, return_all: bool = False, objective: str = "eps", log_snr=log_snr_cosine):
    """continuous_time.py:260-317 (RePaint) on an explicit noise tape.  Draw order: initial x_T; per reverse
    sub-step first the known region's forward noise, then p_step's noise; per forward sub-step one draw."""
    tape = iter(noises)
    nxt = lambda: next(tape).to(device=known.device, dtype=known.dtype)
    B = known.shape[0]
    x_t = nxt()
    out = [x_t]
    steps = torch.linspace(1.0, 0.0, num_steps + 1, device=known.device)[None].repeat_interleave(B, dim=0)
    x_s = x_t
    for i in range(num_steps):
        for j in range(num_resample_steps):
            t, s = steps[:, [i]], steps[:, [i + 1]]
            r = t + torch.linspace(0, 1, jump_length + 1, device=known.device)[None] * (s - t)
            x = x_t
            for k in range(jump_length):  # t -> s
                known_s = q_step_from_x_0(known, r[:, k + 1], nxt(), log_snr)
                unknown_s = p_step_continuous(denoise, x, r[:, k], r[:, k + 1], nxt(), "ddpm", 0.0, objective, log_snr)
                x = mask * known_s + (1 - mask) * unknown_s
            x_s = x
            out.append(x_s)
            if i == num_steps - 1 or j == num_resample_steps - 1:
                x_t = x
                break
            for k in range(jump_length, 0, -1):  # s -> t
                x = q_step(x, r[:, k - 1], r[:, k], nxt(), log_snr)
            x_t = x
    return torch.stack(out) if return_all else x_s


# --------------------------------------------------------------------------------------
# L3 diffusion (discrete time; secondary path, discrete_time.py)
# --------------------------------------------------------------------------------------
def discrete_tables(num_training_steps: int, schedule: str = "linear"):
    """discrete_time.py:12-48,57-78 -- float64 tables cast to float32."""
    T = num_training_steps
    if schedule == "linear":
        s = 1000 / T
        beta = torch.linspace(s * 1e-4, s * 0.02, T, dtype=torch.float64)
    elif schedule == "cosine":
        t = torch.linspace(0, T, T + 1, dtype=torch.float64) / T
        ab = torch.cos((t + 0.008) / 1.008 * math.pi * 0.5) ** 2
        ab = ab / ab[0]
        beta = torch.clip(1 - ab[1:] / ab[:-1], 0, 0.999)
    elif schedule == "sigmoid":
        t = torch.linspace(0, T, T + 1, dtype=torch.float64) / T
        vs, ve = torch.tensor(-3.0).sigmoid(), torch.tensor(3.0).sigmoid()
        ab = (-((t * 6 - 3)).sigmoid() + ve) / (ve - vs)
        ab = ab / ab[0]
        beta = torch.clip(1 - ab[1:] / ab[:-1], 0, 0.999)
    else:
        raise ValueError(schedule)
    alpha_bar = torch.cumprod(1 - beta, dim=0)
    alpha_bar_prev = torch.cat([torch.ones(1, dtype=torch.float64), alpha_bar[:-1]])
    return beta.float(), alpha_bar.float(), alpha_bar_prev.float()


def p_step_discrete(denoise, tables, x_t, steps, noise, mode="ddim", eta=0.0, objective="eps", clip=1.0):
    """discrete_time.py:126-180 with explicit noise (ignored by DDIM when eta == 0, :173)."""
    v4 = lambda t: t[steps][:, None, None, None]
    beta, ab, abp = (v4(t.to(x_t.device)) for t in tables)
    alpha = 1 - beta
    pred = denoise(x_t, steps)
    if objective == "eps":
        x_0 = ab.rsqrt() * x_t - (ab.reciprocal() - 1).sqrt() * pred
    elif objective == "x_0":
        x_0 = pred
    elif objective == "v":
        x_0 = ab.sqrt() * x_t - (1 - ab).sqrt() * pred
    else:
        raise ValueError(objective)
    if clip is not None:
        x_0 = x_0.clamp(-clip, clip)
    nz = noise * (steps != 0).to(noise)[:, None, None, None] if noise is not None else None
    if mode == "ddpm":
        mean = abp.sqrt() * beta / (1 - ab) * x_0 + (1 - abp) * alpha.sqrt() / (1 - ab) * x_t
        var = (beta * (1 - abp) / (1 - ab)).clamp(min=1e-20)
        return mean + (0.5 * var.log()).exp() * nz
    if mode == "ddim":
        var = (1 - abp) / (1 - ab) * (1 - ab / abp)
        std = eta * var.sqrt()
        eps = (x_t - ab.sqrt() * x_0) / (1 - ab).sqrt()
        x_s = abp.sqrt() * x_0 + (1 - abp - std**2).sqrt() * eps
        return x_s + std * nz if eta > 0 else x_s
    raise ValueError(mode)


# --------------------------------------------------------------------------------------
# adjacent post-processing (utils/lidar.py) used by the drop-in scripts
# --------------------------------------------------------------------------------------
def hdl64e_ray_angles(H: int, W: int) -> Tensor:
    """utils/lidar.py:9-20."""
    el = (1 - torch.arange(H) / H) * 28 - 25
    az = (1 - torch.arange(W) / W) * 360 - 180
    el, az = torch.meshgrid(el, az, indexing="ij")
    return torch.stack([el, az])[None].deg2rad()


def lidar_postprocess(x: Tensor, ray_angles: Tensor, min_depth=1.45, max_depth=80.0) -> Tensor:
    """sample_and_save.py:52-57 with utils/lidar.py:49-120 (log_depth): (B,2,H,W) in [-1,1]
    -> (B,5,H,W) = [metric depth, x, y, z, reflectance]."""
    x = (x + 1) / 2
    depth, rflct = x[:, [0]], x[:, [1]]
    metric = torch.exp2(depth * math.log2(max_depth + 1)) - 1
    mask = ((metric > min_depth) & (metric < max_depth)).float()
    metric = metric * mask
    phi, theta = ray_angles[:, [0]].to(x), ray_angles[:, [1]].to(x)
    xyz = torch.cat([metric * phi.cos() * theta.cos(), metric * phi.cos() * theta.sin(),
                     metric * phi.sin()], dim=1)
    xyz = xyz * ((metric > min_depth) & (metric < max_depth)).float()
    return torch.cat([metric, xyz, rflct], dim=1)
