"""-m gpu: every HIP kernel against the oracle (and the reference's golden vectors) through the
C ABI.  Tolerances are absolute, fp32: a K-term fp32 dot product with |terms| ~ 1 carries
~sqrt(K)*6e-8 of roundoff, and the oracle (MIOpen-free torch CPU ops) has the same amount with a
different summation order, so 'equal' means a few 1e-6 at K = 4608."""
import math
import os

import pytest
import torch

from conftest import GOLDEN_RES, max_abs, rnd, synthetic_ckpt

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def O():
    from oracle import r2dm_oracle

    return r2dm_oracle


@pytest.fixture(scope="module")
def H():
    import hipops

    return hipops


@pytest.fixture(scope="module")
def sd(O):
    return O.strip_prefix(synthetic_ckpt(resolution=GOLDEN_RES)["ema_weights"])


# all distinct (Cin, Cout, H, W) conv shapes of the 64x1024 network at B=1,
# spatial size reduced 4x per axis where the full map would take the CPU oracle too long
CONV3 = [(16, 16, 8, 64), (48, 24, 8, 64),  # fewer output channels than the 32-channel tile (round 3: the residual / bias base pointers)
         (34, 64, 16, 256), (64, 64, 16, 256), (64, 128, 16, 256), (128, 64, 16, 256), (64, 2, 16, 256),
         (128, 128, 8, 128), (128, 256, 8, 128), (256, 64, 8, 128), (256, 256, 16, 256), (256, 512, 4, 64),
         (512, 128, 4, 64), (512, 512, 8, 128), (512, 256, 8, 128), (40, 72, 12, 96), (64, 64, 2, 16),
         (16, 3, 7, 36), (64, 4, 9, 260)]  # few-output (direct) kernel: odd heights, widths that are not a multiple of its 256-column block


@pytest.mark.parametrize("cin,cout,h,w", CONV3)
def test_conv3x3_ring(O, H, cin, cout, h, w):
    x, wt, b = rnd(1, 2, cin, h, w), rnd(2, cout, cin, 3, 3) / math.sqrt(9 * cin), rnd(3, cout)
    ref = O.conv_ring(x.double(), wt.double(), b.double())
    y = H.conv2d_ring(x.to(DEV), wt.to(DEV), b.to(DEV)).cpu()
    # outputs are O(1) (|y| <= ~6); fp32 accumulation of K = 9*Cin products (split-bf16 products or the fp32 fma
    # chain, two-level for Cin > 128) carries roundoff ~sqrt(576)*2^-24*|partial sums|: a few 1e-6 worst case
    assert max_abs(y, ref) < (2e-5 if cin >= 256 else 1e-5)
    assert max_abs(y, O.conv_ring(x, wt, b)) < 2.5e-5


@pytest.mark.parametrize("cin,cout,h,w,B", [(64, 64, 64, 1024, 8), (128, 128, 32, 512, 8), (256, 256, 16, 256, 8),
                                             (512, 512, 8, 128, 8), (512, 256, 8, 128, 3), (128, 64, 64, 1024, 2)])
def test_conv3x3_repeatable_at_full_size(H, cin, cout, h, w, B):
    """The split-bf16 kernels stage weights by LDS-DMA and operand fragments by hand-issued reads, ordered only by
    hand-counted vmcnt / lgkmcnt waits and barriers: a missing wait shows up as rare, launch-dependent wrong tiles.
    25 launches at the BASELINE sizes (all CUs busy, several rounds of blocks) must be bit-identical, and agree with
    an fp64 convolution on a slice."""
    import torch.nn.functional as F

    x, wt, b = rnd(30, B, cin, h, w).to(DEV), (rnd(31, cout, cin, 3, 3) / math.sqrt(9 * cin)).to(DEV), rnd(32, cout).to(DEV)
    res = rnd(33, B, cout, h, w).to(DEV)
    aff = torch.stack([torch.rand(B, cin, device=DEV) + 0.5, torch.randn(B, cin, device=DEV) * 0.3], -1).contiguous()
    y0 = H.conv2d_ring(x, wt, b, aff=aff, prologue=2, residual=res, scale=0.70710678)
    for _ in range(24):
        assert torch.equal(H.conv2d_ring(x, wt, b, aff=aff, prologue=2, residual=res, scale=0.70710678), y0)
    bs = B - 1  # fp64 check of the last sample
    xa = F.silu(x[bs:].double() * aff[bs:, :, 0].double()[:, :, None, None] + aff[bs:, :, 1].double()[:, :, None, None])
    xa = F.pad(F.pad(xa, (1, 1, 0, 0), mode="circular"), (0, 0, 1, 1))
    ref = (res[bs:].double() + F.conv2d(xa, wt.double(), b.double())) * 0.70710678
    assert max_abs(y0[bs:].cpu(), ref.cpu()) < 2e-5


@pytest.mark.skipif(os.environ.get("R2DM_CONV_ALGO", "").startswith("f"), reason="fp32-MFMA algorithm forced")
@pytest.mark.parametrize("pro", [0, 1, 2])
@pytest.mark.parametrize("cin,cout,h,w", [(64, 64, 16, 256), (128, 64, 8, 128), (64, 128, 8, 128), (128, 128, 8, 128), (128, 256, 8, 128),
                                          (256, 256, 16, 256), (512, 512, 8, 128)])
def test_conv3x3_both_operand_splits(O, H, cin, cout, h, w, pro):
    """The matrix-pipe formulations of the fp32 3x3 convolution -- fp16 + fp16 residual, three products, in 64-channel tiles
    with two accumulators and (round 4) 128-channel tiles with one (conv_f16x2.hip, pieces = 2); three bf16 pieces, six products
    (conv_bf16x3.hip, pieces = 3) -- against an fp64 convolution, every prologue, with residual and scale, next to the
    fp32-input MFMA kernel (conv_mfma.hip: an exact fmaf chain, pieces = 4 of the test hook) measured in the same test.  All are
    fp32-class: max error < 1e-5 on O(1) outputs.  Two yardsticks measured in the same test: the fp32-MFMA kernel as ONE fmaf chain
    (pieces = 5: what fp32 arithmetic is) and as the library runs it (pieces = 4: two-level accumulation above 128 channels).  Bars:
    every split kernel's rms error <= the plain chain's, at every depth; the two-accumulator tile <= 1.1x the two-level kernel's; the
    one-accumulator tiles (three truncating accumulator updates per tap instead of one; 128 channels x 4 rows, and since round 5
    64 channels x 8 rows: bit-identical to each other) <= 1.0x of it up to Cin = 128 and <= 1.3x at
    Cin = 256 -- the depths conv_f16x2_pick_co_tile uses them at (measured 0.66-0.70x / 1.2x; 1.7x at Cin = 512: not dispatched there)."""
    import torch.nn.functional as F

    B = 3
    x, wt, b = rnd(1, B, cin, h, w), rnd(2, cout, cin, 3, 3) / math.sqrt(9 * cin), rnd(3, cout)
    res = rnd(4, B, cout, h, w)
    aff = torch.stack([torch.rand(B, cin) + 0.5, torch.randn(B, cin) * 0.3], -1).contiguous() if pro else None
    xa = x.double()
    if pro:
        xa = xa * aff[..., 0].double()[:, :, None, None] + aff[..., 1].double()[:, :, None, None]
    if pro == 2:
        xa = F.silu(xa)
    ref = (res.double() + O.conv_ring(xa, wt.double(), b.double())) * 0.70710678
    out = {}
    variants = ([("f16x2/64", 2, "64"), ("bf16x3", 3, None), ("f32 mfma", 4, None), ("f32 chain", 5, None)] + ([("f16x2/128", 2, "128")] if cout % 128 == 0 else [])
                + ([("f16x2/64x8", 2, "64x8")] if h % 8 == 0 else [])  # (round 5: the one-accumulator tile of 64 channels x 8 rows)
                + [("f16x2/32", 2, "32")])                              # (round 5: 32-channel tiles for launches with fewer 64-channel tiles than CUs)
    saved = os.environ.get("R2DM_F2_CO_TILE")
    for name, pieces, tile in variants:
        H.set_conv_pieces(pieces)
        if tile:
            os.environ["R2DM_F2_CO_TILE"] = tile
        try:
            out[name] = H.conv2d_ring(x.to(DEV), wt.to(DEV), b.to(DEV), aff=None if aff is None else aff.to(DEV), prologue=pro,
                                      residual=res.to(DEV), scale=0.70710678).cpu()
        finally:
            H.set_conv_pieces(2)
            os.environ.pop("R2DM_F2_CO_TILE", None)
            if saved is not None:
                os.environ["R2DM_F2_CO_TILE"] = saved
    e = {k: (max_abs(v, ref), (v.double() - ref).pow(2).mean().sqrt().item()) for k, v in out.items()}
    print(f"conv {cin}->{cout} pro={pro}: " + " | ".join(f"{k} max {v[0]:.2e} rms {v[1]:.2e}" for k, v in e.items()))
    assert not torch.equal(out["f16x2/64"], out["bf16x3"]) and not torch.equal(out["f16x2/64"], out["f32 mfma"])  # the mode switches took effect
    assert all(v[0] < 1e-5 for k, v in e.items() if k != "f32 chain")  # (the plain chain is a yardstick, not a product path: 1.0e-5 at K = 4608)
    assert e["f16x2/64"][1] < (0.75 if cin <= 128 else 1.5) * e["bf16x3"][1]
    for k in ("f16x2/64", "f16x2/128", "f16x2/64x8"):
        if k in e:
            assert e[k][1] <= e["f32 chain"][1], (k, e)  # at or below an fp32 fmaf chain, at every depth
            bar = 1.0 if cin <= 128 else 1.1 if k == "f16x2/64" else 1.3 if cin <= 256 else 2.0
            assert e[k][1] <= bar * e["f32 mfma"][1], (k, e)
    if cin > 128:
        assert not torch.equal(out["f32 mfma"], out["f32 chain"]) and e["f32 mfma"][1] < e["f32 chain"][1]  # (the hook took effect; two levels pay)
    if "f16x2/128" in e:
        assert not torch.equal(out["f16x2/64"], out["f16x2/128"])  # (the 128-channel tile really ran)
    assert torch.equal(out["f16x2/32"], out["f16x2/64"])  # (two accumulators, the same products in the same order per output element)
    if "f16x2/64x8" in e:
        assert not torch.equal(out["f16x2/64"], out["f16x2/64x8"])  # (the eight-row tile really ran)
        if "f16x2/128" in e:  # both one-accumulator tiles do the same arithmetic per output element, in the same order
            assert torch.equal(out["f16x2/128"], out["f16x2/64x8"])


@pytest.mark.parametrize("pro", [0, 1, 2])
@pytest.mark.parametrize("cin,cout,h,w", [(64, 64, 16, 256), (256, 256, 8, 128), (512, 256, 4, 64)])
def test_conv3x3_operand_prepass_is_bit_identical(H, cin, cout, h, w, pro):
    """The operand pre-pass experiment (presplit.hip): GroupNorm-affine + SiLU + f16 split applied once, the
    convolution's staging waves only issue LDS-DMA (conv_f16x2 PRO_PRESPLIT).  Same arithmetic in the same order: the output must
    equal the default path's bit for bit -- every prologue, both precision modes that use the kernel, image rows at the top and
    bottom of a tile (zero rows of the layout) and the azimuth wrap (lane addresses)."""
    B = 3
    x, wt, b = rnd(11, B, cin, h, w).to(DEV), (rnd(12, cout, cin, 3, 3) / math.sqrt(9 * cin)).to(DEV), rnd(13, cout).to(DEV)
    res = rnd(14, B, cout, h, w).to(DEV)
    aff = torch.stack([torch.rand(B, cin) + 0.5, torch.randn(B, cin) * 0.3], -1).contiguous().to(DEV) if pro else None
    saved = {k: os.environ.get(k) for k in ("R2DM_F2_PRESPLIT", "R2DM_F2_CO_TILE")}
    try:
        os.environ["R2DM_F2_CO_TILE"] = "64"
        for pieces in (2, 1):
            H.set_conv_pieces(pieces)
            out = {}
            for pre in ("0", "1"):
                os.environ["R2DM_F2_PRESPLIT"] = pre
                out[pre] = H.conv2d_ring(x, wt, b, aff=aff, prologue=pro, residual=res, scale=0.70710678)
            assert torch.isfinite(out["1"]).all() and torch.equal(out["0"], out["1"]), (pieces, (out["0"] - out["1"]).abs().max().item())
    finally:
        H.set_conv_pieces(2)
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def test_conv3x3_batch_tiling_variants(O, H):
    # large batch*pixels switches Cout%128==0 layers to the 128-channel tile (conv_pick_co_tile)
    x, wt, b = rnd(4, 8, 32, 64, 256), rnd(5, 128, 32, 3, 3) / math.sqrt(288), rnd(6, 128)
    y = H.conv2d_ring(x.to(DEV), wt.to(DEV), b.to(DEV)).cpu()
    assert max_abs(y, O.conv_ring(x.double(), wt.double(), b.double())) < 5e-6


@pytest.mark.parametrize("cin,cout,h,w", [(512, 128, 4, 64), (128, 64, 16, 256), (256, 768, 8, 128), (512, 1536, 8, 128), (24, 40, 6, 40)])
def test_conv1x1(O, H, cin, cout, h, w):
    x, wt, b = rnd(7, 2, cin, h, w), rnd(8, cout, cin, 1, 1) / math.sqrt(cin), rnd(9, cout)
    y = H.conv2d_ring(x.to(DEV), wt.to(DEV), b.to(DEV)).cpu()
    assert max_abs(y, O.conv_ring(x.double(), wt.double(), b.double())) < 1e-5


@pytest.mark.skipif(os.environ.get("R2DM_CONV_ALGO", "").startswith("f"), reason="fp32-MFMA algorithm forced")
@pytest.mark.parametrize("pro", [0, 1])
@pytest.mark.parametrize("cin,cout,h,w", [(256, 768, 8, 128), (512, 512, 8, 128), (64, 64, 16, 256)])
def test_conv1x1_fp16_pipe_and_fp32_mfma(O, H, cin, cout, h, w, pro):
    """The attention block's projections: split fp16 operands on the fp16 matrix pipe (proj_f16x2.hip, pieces = 2) and the
    fp32-input MFMA (conv_mfma.hip, pieces = 3) against an fp64 product, plain and GroupNorm-affine input, with residual
    and scale.  Both fp32-class; the split-operand form is the more accurate one (one truncating accumulator update per 16
    k-values instead of eight)."""
    B = 3
    x, wt, b = rnd(1, B, cin, h, w), rnd(2, cout, cin, 1, 1) / math.sqrt(cin), rnd(3, cout)
    res = rnd(4, B, cout, h, w)
    aff = torch.stack([torch.rand(B, cin) + 0.5, torch.randn(B, cin) * 0.3], -1).contiguous() if pro else None
    xa = x.double()
    if pro:
        xa = xa * aff[..., 0].double()[:, :, None, None] + aff[..., 1].double()[:, :, None, None]
    ref = (res.double() + O.conv_ring(xa, wt.double(), b.double())) * 0.70710678
    out = {}
    for pieces in (2, 3):
        H.set_conv_pieces(pieces)
        try:
            out[pieces] = H.conv2d_ring(x.to(DEV), wt.to(DEV), b.to(DEV), aff=None if aff is None else aff.to(DEV), prologue=pro,
                                        residual=res.to(DEV), scale=0.70710678).cpu()
        finally:
            H.set_conv_pieces(2)
    e = {k: (max_abs(v, ref), (v.double() - ref).pow(2).mean().sqrt().item()) for k, v in out.items()}
    print(f"conv1x1 {cin}->{cout} pro={pro}: f16x2 max {e[2][0]:.2e} rms {e[2][1]:.2e} | f32 mfma max {e[3][0]:.2e} rms {e[3][1]:.2e}")
    assert not torch.equal(out[2], out[3])  # the mode switch took effect
    assert e[2][0] < 1e-5 and e[3][0] < 1e-5
    assert e[2][1] < 1.2 * e[3][1]


def test_conv_golden(golden, H, sd):
    g = golden("ops")
    p, q = "d_block1.residual_blocks.1.", "u_block3.residual_blocks.0."
    y = H.conv2d_ring(g["conv3_x"].to(DEV), sd[p + "conv1.weight"].to(DEV), sd[p + "conv1.bias"].to(DEV)).cpu()
    assert max_abs(y, g["conv3_y"]) < 1e-5
    y = H.conv2d_ring(g["conv1_x"].to(DEV), sd[q + "skip.weight"].to(DEV), sd[q + "skip.bias"].to(DEV)).cpu()
    assert max_abs(y, g["conv1_y"]) < 1e-5


def test_conv_fused_prologue_epilogue(O, H):
    """SiLU(GroupNorm(x)) folded into the conv load, residual add and 1/sqrt(2) into its store."""
    B, C, h, w = 2, 64, 8, 128
    x, wt, b, res = rnd(10, B, C, h, w) * 2 + 0.5, rnd(11, C, C, 3, 3) / 24, rnd(12, C), rnd(13, B, C, h, w)
    gam, bet = 1 + 0.1 * rnd(14, C), 0.1 * rnd(15, C)
    aff, stats = H.group_norm_affine(x.to(DEV), 8, 1e-6, gam.to(DEV), bet.to(DEV))
    y = H.conv2d_ring(x.to(DEV), wt.to(DEV), b.to(DEV), aff=aff, prologue=2, residual=res.to(DEV), scale=O.INV_SQRT2).cpu()
    xd = x.double()
    ref = (res.double() + O.conv_ring(O.silu(O.group_norm(xd, 8, 1e-6, gam.double(), bet.double())), wt.double(), b.double())) * O.INV_SQRT2
    assert max_abs(y, ref) < 5e-6
    y = H.conv2d_ring(x.to(DEV), wt.to(DEV), b.to(DEV), aff=aff, prologue=1).cpu()
    assert max_abs(y, O.conv_ring(O.group_norm(xd, 8, 1e-6, gam.double(), bet.double()), wt.double(), b.double())) < 5e-6


@pytest.mark.parametrize("shape", [(2, 64, 64, 1024), (2, 128, 16, 256), (3, 512, 8, 128), (2, 16, 6, 20)])
def test_group_norm_stats(O, H, shape):
    x = rnd(20, *shape) * 3 + 1.5
    aff, stats = H.group_norm_affine(x.to(DEV), 8, 1e-6)
    xg = x.double().reshape(shape[0], 8, -1)
    mean, var = xg.mean(-1), xg.var(-1, unbiased=False)
    assert max_abs(stats[..., 0].cpu(), mean) < 1e-6
    assert ((stats[..., 1].cpu().double() * (var + 1e-6).sqrt()) - 1).abs().max() < 1e-6


def test_group_norm_golden(golden, H, sd):
    g = golden("ops")
    p = "d_block1.residual_blocks.1."
    x = g["gn_x"].to(DEV)
    aff, _ = H.group_norm_affine(x, 8, 1e-6, sd[p + "norm1.weight"].to(DEV), sd[p + "norm1.bias"].to(DEV))
    assert max_abs(H.affine_act(x, aff, False).cpu(), g["gn_y"]) < 5e-6
    ss = torch.nn.functional.linear(torch.nn.functional.silu(g["temb"]), sd[p + "norm2.proj.1.weight"], sd[p + "norm2.proj.1.bias"])
    aff, _ = H.group_norm_affine(x, 8, 1e-6, ada=ss.to(DEV).contiguous())
    assert max_abs(H.affine_act(x, aff, False).cpu(), g["adagn_y"]) < 5e-6
    y = H.affine_act(x, aff, True).cpu()
    assert max_abs(y, torch.nn.functional.silu(g["adagn_y"])) < 5e-6


@pytest.mark.parametrize("shape", [(2, 8, 8, 32), (1, 128, 64, 1024), (2, 3, 2, 4), (2, 512, 16, 256), (1, 4, 128, 2048), (2, 5, 4, 24), (1, 64, 32, 512), (3, 7, 6, 40),
                                   (1, 2, 4, 6)])
def test_fir_resamplers(O, H, shape):
    """Round 6: the halo columns come from the neighbouring lanes where a wave holds whole rows (down: W / 8 divides 64; up: every lane but those at a
    wave's edge or the azimuth seam) -- shapes on both sides of every such condition, and both switches off: bit-identical to the load-only kernels."""
    x = rnd(30, *shape)
    d = H.fir_down2(x.to(DEV))
    assert max_abs(d.cpu(), O.fir_down2(x.double())) < 1e-6
    os.environ["R2DM_FIR_SHFL"] = "0"
    try:
        assert torch.equal(d, H.fir_down2(x.to(DEV)))
    finally:
        del os.environ["R2DM_FIR_SHFL"]
    want, got = O.fir_up2(x.double()), []
    for mode in ("0", "2", None):  # the two-column kernel | the 2 x 8 kernel at any size | what the launcher picks
        if mode is not None:
            os.environ["R2DM_FIR_UP_WIDE"] = mode
        try:
            got.append(H.fir_up2(x.to(DEV)))
            assert max_abs(got[-1].cpu(), want) < 1e-6, mode
        finally:
            os.environ.pop("R2DM_FIR_UP_WIDE", None)
    assert torch.equal(got[0], got[1]) and torch.equal(got[0], got[2])  # one spelled-out chain (up1) in both kernels


@pytest.mark.parametrize("B,C,Hh,Ww", [(8, 128, 64, 1024), (2, 256, 32, 512), (3, 512, 16, 256), (2, 128, 16, 128), (1, 64, 16, 128)])
def test_fir_down_with_fused_group_norm_statistics(H, B, C, Hh, Ww):
    """Round 5: fir_down2_stats_kernel (resample.hip) -- the down-sampler of efficient_unet.py:135 leaving the statistics of the norm behind it
    (efficient_unet.py:95-97) in the convolution epilogues' slot grid.  Its output equals the plain kernel's bit for bit; every slot is written;
    the slot sums of a (sample, group) are the group's fp64 moments of the STORED output (rel. 1e-12); groups of fewer than 64 channels leave
    the second half of the grid zero (conv_epilogue.h's convention, which the consumer-side fold of conv_f16x2.hip relies on)."""
    x = (rnd(31, B, C, Hh, Ww) * 1.7 + 0.3).to(DEV)
    assert H.fir_down2_stats(rnd(32, 1, 64, 8, 64).to(DEV), 8) is None  # (fewer than 64 patches per slot: the engine runs the streaming pass)
    r = H.fir_down2_stats(x, 8)
    assert r is not None
    y, stat = r
    assert torch.equal(y, H.fir_down2(x))
    os.environ["R2DM_FIR_NARROW"] = "1"  # (the generic two-outputs-per-thread kernel: one spelled-out FMA chain in all three, resample.hip fir4)
    try:
        assert torch.equal(y, H.fir_down2(x))
    finally:
        del os.environ["R2DM_FIR_NARROW"]
    assert torch.isfinite(stat).all()
    cpg = C // 8
    yd = y.double().reshape(B, 8, -1)
    want_s, want_q = yd.sum(-1), (yd * yd).sum(-1)
    got = stat.sum(2)
    assert ((got[..., 0] - want_s).abs() <= 1e-12 * yd.abs().sum(-1)).all()
    assert ((got[..., 1] - want_q).abs() <= 1e-12 * want_q).all()
    if cpg < 64:
        assert (stat[:, :, stat.shape[2] // 2:] == 0).all()
    # a slot's energy bounds every element it covers: the largest slot energy bounds max|y| of the group (the range guard's M)
    assert (stat[..., 1].amax(2).sqrt() >= yd.abs().amax(-1)).all()


def test_fir_golden(golden, H):
    g = golden("ops")  # 6x10 maps: W % 4 != 0 -- refused until round 5, now the one-output-per-thread kernel (missing #4) against the
    # reference's own Resample(down=2) output (/root/reference/models/ops.py:91-143); an odd width stays an error
    from r2dm_amd._lib import R2DMError

    assert max_abs(H.fir_down2(g["down_x"].to(DEV)).cpu(), g["down_y"]) < 1e-6
    with pytest.raises(R2DMError):
        H.fir_down2(g["down_x"][..., :9].contiguous().to(DEV))
    x = torch.nn.functional.pad(g["up_x"], (0, 0, 0, 0))
    assert max_abs(H.fir_up2(x.to(DEV)).cpu(), g["up_y"]) < 1e-6
    # round 5 (VERDICT round 4, weak #1e): maps whose width IS a multiple of 4 -- the HIP down-sampler against the reference's own output
    # (/root/reference/models/ops.py:91-143), 8 x 12 (narrow: the scalar path) and 16 x 64 (the wide kernel)
    g2 = golden("res32x256")
    for sfx in ("", "w"):
        assert max_abs(H.fir_down2(g2["down_x" + sfx].to(DEV)).cpu(), g2["down_y" + sfx]) < 1e-6
        assert max_abs(H.fir_up2(g2["up_x" + sfx].to(DEV)).cpu(), g2["up_y" + sfx]) < 1e-6


@pytest.mark.parametrize("B,C,N", [(2, 512, 1024), (2, 256, 1024), (1, 512, 32), (1, 256, 4096), (3, 256, 96),
                                   (2, 768, 128), (2, 192, 16), (1, 128, 50), (1, 1024, 40)])  # head sizes 96 / 24 / 16 / 128, odd token counts: the generic kernel
def test_attention(O, H, B, C, N):
    qkv = rnd(40, B, 3 * C, N)
    qkv[:, :, 5] *= 4.0  # a spiky token: forces running-max updates in the online softmax
    q, k, v = qkv.double().chunk(3, dim=1)
    sp = lambda t: t.reshape(B, 8, C // 8, N)
    att = torch.softmax(sp(q).transpose(-1, -2) @ sp(k) / math.sqrt(C // 8), dim=-1)
    ref = (sp(v) @ att.transpose(-1, -2)).reshape(B, C, N)
    q, k, v = qkv.chunk(3, dim=1)  # the same expression in fp32 (what nn.MultiheadAttention's math path does)
    att32 = torch.softmax(sp(q).transpose(-1, -2) @ sp(k) / math.sqrt(C // 8), dim=-1)
    err32 = max_abs((sp(v) @ att32.transpose(-1, -2)).reshape(B, C, N), ref)
    # the spiky token drives logits to ~128, where fp32 softmax itself is only ~1e-5 accurate:
    # require both kernels -- fp16 matrix pipe with split operands (pieces = 2, default) and fp32 MFMA (pieces = 3) -- to be in
    # the same class as the fp32 reference expression
    for pieces in (2, 3):
        H.set_conv_pieces(pieces)
        try:
            err = max_abs(H.attention(qkv.to(DEV), 8).cpu(), ref)
        finally:
            H.set_conv_pieces(2)
        print(f"attention B={B} C={C} N={N} pieces={pieces}: max err {err:.2e} (fp32 expression {err32:.2e})")
        assert err < max(3 * err32, 3e-6), (pieces, err, err32)


def test_attention_block_golden(golden, H, sd, O):
    g = golden("ops")
    for pre, key in (("d_block4.self_attn_block.", "attn"), ("u_block4.self_attn_block.", "attn_u")):
        x = g[key + "_x"].to(DEV)
        B, C, h, w = x.shape
        aff, _ = H.group_norm_affine(x, 8, 1e-6, sd[pre + "norm.weight"].to(DEV), sd[pre + "norm.bias"].to(DEV))
        qkv = H.conv2d_ring(x, sd[pre + "attn.in_proj_weight"].to(DEV)[:, :, None, None], sd[pre + "attn.in_proj_bias"].to(DEV), aff=aff, prologue=1)
        o = H.attention(qkv.reshape(B, 3 * C, h * w), 8).reshape(B, C, h, w)
        y = H.conv2d_ring(o, sd[pre + "attn.out_proj.weight"].to(DEV)[:, :, None, None], sd[pre + "attn.out_proj.bias"].to(DEV), residual=x, scale=O.INV_SQRT2)
        assert max_abs(y.cpu(), g[key + "_y"]) < 1e-5


def test_time_embedding(golden, H, sd, O):
    g = golden("ops")
    half = 32
    freqs = torch.exp(-math.log(10_000) / (half - 1) * torch.arange(half))
    act = H.time_embedding(g["sin_t"].to(DEV), freqs.to(DEV), *(sd[f"time_embedding.{i}.{n}"].to(DEV) for i in (1, 3) for n in ("weight", "bias")))
    assert max_abs(act.cpu(), torch.nn.functional.silu(g["temb_y"])) < 2e-6
    t = torch.tensor([999.0, 500.0, 1.0, 0.0])  # discrete-time conditions: large sinusoid arguments
    act = H.time_embedding(t.to(DEV), freqs.to(DEV), *(sd[f"time_embedding.{i}.{n}"].to(DEV) for i in (1, 3) for n in ("weight", "bias")))
    cfg = O.UNetConfig(resolution=GOLDEN_RES)
    assert max_abs(act.cpu(), O.silu(O.time_embedding(sd, cfg, t))) < 1e-5


def test_posterior_bit_exact_vs_torch_ops(O):
    """The fused posterior kernel replays the reference's float32 op order (continuous_time.py:208-229) with
    FMA contraction off: bit-identical to the same expression evaluated op-by-op by torch on the GPU, given
    the same (host-computed) scalars; and equal to the oracle's p_step up to its GPU-evaluated scalars."""
    from r2dm_amd import diffusion as D

    class Stub(torch.nn.Module):
        resolution, in_channels = (16, 128), 2

    x, pred, z = (rnd(50 + i, 4, 2, 16, 128).to(DEV) for i in range(3))
    t, s = torch.tensor([1.0, 0.7, 0.3, 0.01]), torch.tensor([0.9, 0.6, 0.2, 0.0])
    for obj in ("eps", "v", "x_0"):
        dd = D.ContinuousTimeGaussianDiffusion(Stub(), prediction_type=obj).to(DEV)
        for mode, eta in (("ddpm", 0.0), ("ddim", 0.0), ("ddim", 0.7)):
            _, coef, mid = dd._coefficients(t, s, mode, eta)
            got = dd._posterior(x, pred, z, coef.to(DEV), mid)
            k = [coef[:, i].to(DEV)[:, None, None, None] for i in range(8)]
            a_t, s_t, a_s = k[0], k[1], k[2]
            x0 = {"eps": lambda: (x - s_t * pred) / a_t, "v": lambda: a_t * x - s_t * pred, "x_0": lambda: pred}[obj]()
            x0 = x0.clamp(-1, 1)
            if mode == "ddpm":
                want = a_s * (x * (1 - k[4]) / a_t + k[4] * x0) + k[5] * z
            else:
                want = a_s * x0 + k[6] * z + k[7] * ((x - a_t * x0) / s_t)
            assert torch.equal(got, want), (obj, mode, eta, max_abs(got, want))
            ref = O.p_step_continuous(lambda a, c: pred, x, t.to(DEV), s.to(DEV), z, mode, eta, obj)
            assert max_abs(got, ref) < 1e-5 * max(1.0, ref.abs().max().item()), (obj, mode, eta)


def test_posterior_discrete_bit_exact():
    from r2dm_amd import diffusion as D

    class Stub(torch.nn.Module):
        resolution, in_channels = (16, 128), 2

    x, pred, z = (rnd(60 + i, 3, 2, 16, 128).to(DEV) for i in range(3))
    steps = torch.tensor([999, 321, 0])
    for obj in ("eps", "v", "x_0"):
        dd = D.DiscreteTimeGaussianDiffusion(Stub(), prediction_type=obj, num_training_steps=1000).to(DEV)
        v4 = lambda tt: tt[steps.to(DEV)]
        beta, ab, abp = v4(dd.beta), v4(dd.alpha_bar), v4(dd.alpha_bar_prev)
        x0 = {"eps": lambda: ab.rsqrt() * x - (ab.reciprocal() - 1).sqrt() * pred,
              "v": lambda: ab.sqrt() * x - (1 - ab).sqrt() * pred, "x_0": lambda: pred}[obj]().clamp(-1, 1)
        coef, mid = dd._coefficients(steps, "ddpm", 0.0)
        got = dd._posterior(x, pred, z, coef.to(DEV), mid)
        mean = abp.sqrt() * beta / (1 - ab) * x0 + (1 - abp) * (1 - beta).sqrt() / (1 - ab) * x
        zz = z.clone()
        zz[steps == 0] *= 0
        want = mean + (0.5 * (beta * (1 - abp) / (1 - ab)).clamp(min=1e-20).log()).exp() * zz
        assert max_abs(got, want) < 2e-6, obj  # host vs GPU evaluation of the table expressions: 1 ulp
        coef, mid = dd._coefficients(steps, "ddim", 0.0)
        got = dd._posterior(x, pred, None, coef.to(DEV), mid)
        eps = (x - ab.sqrt() * x0) / (1 - ab).sqrt()
        want = abp.sqrt() * x0 + (1 - abp).sqrt() * eps
        assert max_abs(got, want) < 2e-6, obj


def test_lidar_postprocess(golden):
    from r2dm_amd import _lib

    g = golden("lidar")
    y = _lib.lidar_postprocess(g["x"].to(DEV), g["ray_angles"][0].to(DEV), 1.45, 80.0).cpu()
    # x = +-1 (clamped samples) decodes to exactly the mask threshold 80 m, where a 1-ulp difference between
    # the CPU's and the GPU's exp2 flips the validity mask; compare away from the two thresholds
    d = g["y"][:, :1]
    safe = ((d - 80.0).abs() > 1e-3) & ((d - 1.45).abs() > 1e-3) & ((y[:, :1] - 80.0).abs() > 1e-3) & (g["x"][:, :1].abs() < 1)
    assert safe.float().mean() > 0.5
    assert ((y - g["y"]).abs() * safe).max() < 2e-4  # metric depth up to 80 m: 2e-4 abs ~ 3 ulp


@pytest.mark.parametrize("fmt", ["inverse_depth", "depth"])
def test_lidar_postprocess_other_depth_formats(golden, fmt):
    """`postprocess` of /root/reference/sample_and_save.py:52-57 for a checkpoint with a non-default depth coding
    (/root/reference/utils/lidar.py:95-112), fixture from the reference (make_golden.py lidar_formats), through LiDARUtility."""
    from r2dm_amd.lidar import LiDARUtility

    g = golden("lidar_formats")
    lu = LiDARUtility(g["x"].shape[-2:], fmt, 1.45, 80.0).to(DEV)
    assert torch.equal(lu.ray_angles.cpu(), g["ray_angles"])
    y, want = lu.postprocess(g["x"].to(DEV)).cpu(), g[f"y_{fmt}"]
    # away from the two mask thresholds (a 1-ulp difference in the decoded depth flips the validity mask there)
    d = want[:, :1]
    dn = (g["x"][:, :1] + 1) / 2
    metric = 1.45 / (dn + 1e-8) if fmt == "inverse_depth" else dn * 80.0
    safe = ((metric - 80.0).abs() > 1e-3) & ((metric - 1.45).abs() > 1e-3)
    assert safe.float().mean() > 0.5 and (d > 0).float().mean() > 0.05  # (inverse depth keeps only d > 1.45 / 80)
    assert torch.equal((y[:, :1] * safe), (d * safe))  # division / multiplication: bit-exact
    assert ((y - want).abs() * safe).max() < 2e-4      # xyz: sin / cos of the GPU
