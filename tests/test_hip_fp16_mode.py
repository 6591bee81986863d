"""-m gpu: `precision="fp16"` -- the reduced-precision bulk mode.

What the reference does: its bulk sampler runs the whole denoiser under fp16 autocast (/root/reference/sample_and_save.py:70,
utils/option.py:49 `mixed_precision = "fp16"`): convolutions and matmuls take fp16 operands and accumulate in fp32.
What this mode does: the SAME kernels as the parity mode with the fp16 piece alone -- one `v_mfma_f32_32x32x16_f16` per 16
k-values instead of three -- i.e. operands rounded to fp16 (11 significant bits; weights after the packers' power-of-two
scale, activations after GroupNorm + SiLU), fp32 accumulation, fp32 tensors in HBM, fp32 GroupNorm statistics / softmax /
posterior update.  It is NOT a parity mode: its tolerance class is stated and asserted here, separately.

Tolerance class.  RNE to fp16 has unit roundoff 2^-11; an operand's relative error is ~uniform, rms 2^-11/sqrt(3) = 2.8e-4
... 2^-12/sqrt(3) (mantissa dependent), two operands per product and K random-sign products give a relative rms error of a
convolution output of ~3e-4 independent of K.  Against an emulation that rounds the operands to fp16 in torch and multiplies
in fp64, the kernel must agree to fp32-accumulation accuracy -- that pins the mode to 'fp16 operands, nothing else lost'."""
import math
import os

import pytest
import torch

from conftest import max_abs, rnd, synthetic_ckpt

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("R2DM_CONV_ALGO", "").startswith("f"), reason="fp32-MFMA algorithm forced")]
DEV = "cuda"


@pytest.fixture(scope="module")
def O():
    from oracle import r2dm_oracle

    return r2dm_oracle


@pytest.fixture(scope="module")
def H():
    import hipops

    return hipops


def rel_rms(a, ref):
    return ((a.double() - ref.double()).pow(2).mean().sqrt() / ref.double().pow(2).mean().sqrt()).item()


def f16_round_weights(w):
    """What the packers store: RNE_f16(w * s) / s with s the power of two that brings max|w| into [2^9, 2^10)."""
    e = math.floor(math.log2(w.abs().max().item()))
    s = 2.0 ** (9 - e)
    return (w.double() * s).half().double() / s


@pytest.mark.parametrize("k,cin,cout,h,w", [(3, 64, 64, 16, 256), (3, 128, 256, 8, 128), (3, 512, 512, 8, 128), (1, 256, 768, 8, 128), (1, 512, 512, 8, 128)])
def test_conv_one_product_is_exactly_fp16_operands(O, H, k, cin, cout, h, w):
    x, wt, b = rnd(1, 2, cin, h, w), rnd(2, cout, cin, k, k) / math.sqrt(k * k * cin), rnd(3, cout)
    res = rnd(4, 2, cout, h, w)
    truth = (res.double() + O.conv_ring(x.double(), wt.double(), b.double())) * 0.70710678
    emul = (res.double() + O.conv_ring(x.half().double(), f16_round_weights(wt), b.double())) * 0.70710678
    H.set_conv_pieces(1)
    try:
        y = H.conv2d_ring(x.to(DEV), wt.to(DEV), b.to(DEV), residual=res.to(DEV), scale=0.70710678).cpu()
    finally:
        H.set_conv_pieces(2)
    e_truth, e_emul = rel_rms(y, truth), max_abs(y, emul)
    print(f"fp16 mode k={k} {cin}->{cout}: rel rms vs fp64 {e_truth:.2e}; max |y - fp16-operand emulation| {e_emul:.2e}")
    assert e_emul < 1e-5            # fp32 accumulation of exactly the emulated products
    assert 3e-5 < e_truth < 6e-4    # ... and really the reduced mode (the parity mode sits at ~2e-7)


@pytest.mark.parametrize("cin,cout,h,w", [(256, 256, 8, 128), (128, 128, 16, 256)])
def test_conv_one_product_tiles_are_bit_identical(H, cin, cout, h, w):
    """Round 5: every tile of conv_f16x2 in the one-product mode -- 64 x 4, 128 x 4, 64 x 8 and the 32-channel tile u_block4 runs on at batch 8
    (whose one-plane weight stage is three DMA pieces for four staging waves) -- computes one product per MAC into one accumulator in the same
    order: bit-identical outputs, with GroupNorm + SiLU prologue, residual and scale."""
    import os

    x, wt, b = rnd(1, 3, cin, h, w), rnd(2, cout, cin, 3, 3) / math.sqrt(9 * cin), rnd(3, cout)
    res = rnd(4, 3, cout, h, w)
    aff = torch.stack([torch.rand(3, cin) + 0.5, torch.randn(3, cin) * 0.3], -1).contiguous()
    outs = {}
    saved = os.environ.get("R2DM_F2_CO_TILE")
    H.set_conv_pieces(1)
    try:
        for tile in ("64", "32", "128", "64x8"):
            os.environ["R2DM_F2_CO_TILE"] = tile
            outs[tile] = H.conv2d_ring(x.to(DEV), wt.to(DEV), b.to(DEV), aff=aff.to(DEV), prologue=2, residual=res.to(DEV), scale=0.70710678).cpu()
    finally:
        H.set_conv_pieces(2)
        os.environ.pop("R2DM_F2_CO_TILE", None)
        if saved is not None:
            os.environ["R2DM_F2_CO_TILE"] = saved
    for tile in ("32", "128", "64x8"):
        assert torch.equal(outs[tile], outs["64"]), tile


@pytest.mark.parametrize("tile", ["64", "32", "128", "64x8"])
@pytest.mark.parametrize("cin,cout,h,w", [(128, 128, 16, 256), (64, 64, 32, 512)])
def test_conv_fp16_storage_is_exact(H, cin, cout, h, w, tile):
    """Round 5: activations stored as fp16 in the one-plane mode, as the reference's autocast stores its convolution
    outputs.  Storage is the ONLY change: on fp16-representable inputs a launch reading fp16 gives bit for bit what the same launch reading the
    same values as fp32 gives, and a launch writing fp16 gives exactly RNE_f16 of the fp32 launch's output -- every tile, with GroupNorm +
    SiLU prologue, fp16 residual and scale."""
    import os

    B = 3
    x, wt, b = rnd(1, B, cin, h, w).half().float(), rnd(2, cout, cin, 3, 3) / math.sqrt(9 * cin), rnd(3, cout)
    res = rnd(4, B, cout, h, w).half().float()
    aff = torch.stack([torch.rand(B, cin) + 0.5, torch.randn(B, cin) * 0.3], -1).contiguous()
    saved = {k: os.environ.get(k) for k in ("R2DM_F2_CO_TILE", "R2DM_TEST_IO16")}
    H.set_conv_pieces(1)
    out = {}
    try:
        os.environ["R2DM_F2_CO_TILE"] = tile
        for io in (0, 1, 2, 3):
            os.environ["R2DM_TEST_IO16"] = str(io)
            out[io] = H.conv2d_ring(x.to(DEV), wt.to(DEV), b.to(DEV), aff=aff.to(DEV), prologue=2, residual=res.to(DEV), scale=0.70710678, io16=io).cpu()
    finally:
        H.set_conv_pieces(2)
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    assert out[0].dtype == torch.float32 and out[3].dtype == torch.float16
    assert torch.equal(out[1], out[0])                       # fp16 input: the same values, the same arithmetic
    assert torch.equal(out[2], out[0].half()) and torch.equal(out[3], out[0].half())  # fp16 output: the fp32 result, rounded once


class _io16:
    """R2DM_TEST_IO16 for the duration of a block (the single-kernel C entries read it per call)."""

    def __init__(self, v):
        self.v = str(v)

    def __enter__(self):
        import os
        self.saved = os.environ.get("R2DM_TEST_IO16")
        os.environ["R2DM_TEST_IO16"] = self.v

    def __exit__(self, *a):
        import os
        os.environ.pop("R2DM_TEST_IO16", None)
        if self.saved is not None:
            os.environ["R2DM_TEST_IO16"] = self.saved


@pytest.mark.parametrize("C,h,w,groups", [(128, 64, 1024, 8), (256, 32, 512, 8), (16, 8, 64, 0)])
def test_fir_resamplers_fp16_storage_is_exact(H, C, h, w, groups):
    """Round 6: the activations of the two full-resolution levels are stored as fp16 in the one-plane mode -- the FIR resamplers'
    second I/O type.  Storage is the ONLY change: on fp16-representable inputs a launch reading fp16 equals the launch reading the same values as fp32 bit
    for bit, a launch writing fp16 equals RNE_f16 of the fp32 launch (both wave-exchange modes of the down-sampler, both up-sampler kernels), and the fused
    GroupNorm statistics are those of the STORED values."""
    B = 2
    x = (rnd(5, B, C, h, w) * 1.3 + 0.2).half().float().to(DEV)
    d32, u32 = H.fir_down2(x), H.fir_up2(x)
    with _io16(1):
        assert torch.equal(H.fir_down2(x, io16=1), d32)            # fp16 in, fp32 out (level 2 -> 3)
    with _io16(3):
        assert torch.equal(H.fir_down2(x, io16=3), d32.half())     # fp16 in, fp16 out (level 1 -> 2)
        assert torch.equal(H.fir_up2(x, io16=3), u32.half())       # (level 2 -> 1)
    with _io16(2):
        assert torch.equal(H.fir_up2(x, io16=2), u32.half())       # fp32 in, fp16 out (level 3 -> 2)
    if groups:
        with _io16(3):
            y, stat = H.fir_down2_stats(x, groups, io16=3)
        assert torch.equal(y, d32.half())
        yd = y.double().reshape(B, groups, -1)
        got = stat.sum(2)
        assert ((got[..., 0] - yd.sum(-1)).abs() <= 1e-12 * yd.abs().sum(-1)).all() and ((got[..., 1] - (yd * yd).sum(-1)).abs() <= 1e-12 * (yd * yd).sum(-1)).all()
        with _io16(1):
            y1, stat1 = H.fir_down2_stats(x, groups, io16=1)
        y0, stat0 = H.fir_down2_stats(x, groups)
        assert torch.equal(y1, y0) and torch.equal(stat1, stat0)


def test_in_out_and_skip_convolutions_fp16_storage_is_exact(H):
    """... and the other kernels of those levels: in_conv (few inputs: fp16 OUTPUT + statistics of the stored values through the engine), out_conv (fp16
    INPUT) and the fp16-operand 1 x 1 skip convolution (fp16 in and out).  Same contract as the FIR kernels and conv_f16x2's IOM template."""
    B, h, w = 2, 16, 256
    H.set_conv_pieces(1)
    try:
        # in_conv: 2 -> 64 (+ a residual map, the engine's constant coordinate term)
        x, wt, b = rnd(1, B, 2, h, w), rnd(2, 64, 2, 3, 3) / math.sqrt(18), rnd(3, 64)
        with _io16(4):  # (bit 2: the engine's few-input kernel, fp32 in and out -- the twin the fp16 launch is compared with)
            y32 = H.conv2d_ring(x.to(DEV), wt.to(DEV), b.to(DEV))
        with _io16(2):
            y16 = H.conv2d_ring(x.to(DEV), wt.to(DEV), b.to(DEV), io16=2)
        assert y16.dtype == torch.float16 and torch.equal(y16, y32.half())
        # out_conv: 64 -> 2
        x, wt, b = rnd(4, B, 64, h, w).half().float(), rnd(5, 2, 64, 3, 3) / math.sqrt(9 * 64), rnd(6, 2)
        y32 = H.conv2d_ring(x.to(DEV), wt.to(DEV), b.to(DEV))
        with _io16(1):
            y1 = H.conv2d_ring(x.to(DEV), wt.to(DEV), b.to(DEV), io16=1)
        assert y1.dtype == torch.float32 and torch.equal(y1, y32)
        # skip convolution: 128 -> 64, 1 x 1, raw input
        x, wt, b = rnd(7, B, 128, h, w).half().float(), rnd(8, 64, 128, 1, 1) / math.sqrt(128), rnd(9, 64)
        y32 = H.conv2d_ring(x.to(DEV), wt.to(DEV), b.to(DEV))
        with _io16(3):
            y3 = H.conv2d_ring(x.to(DEV), wt.to(DEV), b.to(DEV), io16=3)
        assert y3.dtype == torch.float16 and torch.equal(y3, y32.half())
    finally:
        H.set_conv_pieces(2)


def test_unet_fp16_storage_levels(O):
    """The whole denoiser with every activation of levels 1 and 2 stored as fp16 (default in the one-plane mode) against the same mode storing fp16 only
    between a residual block's two convolutions (R2DM_FP16_STORAGE=1, round 5) and fp32 everywhere (0): all three inside the mode's tolerance class against
    the fp64 oracle, and within a few fp16 roundings of each other."""
    import os

    import r2dm_amd

    ck = synthetic_ckpt()
    x, c = rnd(70, 2, 2, 64, 1024).to(DEV), torch.tensor([-3.0, 2.0], device=DEV)
    sd = {k: v.double().to(DEV) for k, v in O.strip_prefix(ck["ema_weights"]).items()}
    truth = O.unet_forward(sd, O.UNetConfig(), x.double(), c.double()).cpu()
    saved = os.environ.get("R2DM_FP16_STORAGE")
    ys = {}
    try:
        for lvl in ("2", "1", "0"):
            os.environ["R2DM_FP16_STORAGE"] = lvl
            m, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=2, precision="fp16")
            ys[lvl] = m.model(x, c).cpu()
            assert torch.equal(ys[lvl], m.model(x, c).cpu())
            del m
    finally:
        os.environ.pop("R2DM_FP16_STORAGE", None)
        if saved is not None:
            os.environ["R2DM_FP16_STORAGE"] = saved
    e = {k: rel_rms(v, truth) for k, v in ys.items()}
    print(f"fp16 mode vs fp64, storage levels 2 / 1 / 0: rel rms {e['2']:.2e} {e['1']:.2e} {e['0']:.2e}; level 2 vs level 0: {rel_rms(ys['2'], ys['0'].double()):.2e}")
    assert all(2e-4 < v < 3e-3 for v in e.values())
    assert not torch.equal(ys["2"], ys["1"]) and rel_rms(ys["2"], ys["0"].double()) < 2e-3


@pytest.mark.parametrize("pro", [1, 2])
def test_conv_one_product_with_fused_prologue(O, H, pro):
    import torch.nn.functional as F

    cin, cout, h, w = 128, 128, 16, 256
    x, wt, b = rnd(1, 2, cin, h, w), rnd(2, cout, cin, 3, 3) / math.sqrt(9 * cin), rnd(3, cout)
    aff = torch.stack([torch.rand(2, cin) + 0.5, torch.randn(2, cin) * 0.3], -1).contiguous()
    xa = x.double() * aff[..., 0].double()[:, :, None, None] + aff[..., 1].double()[:, :, None, None]
    if pro == 2:
        xa = F.silu(xa)
    truth = O.conv_ring(xa, wt.double(), b.double())
    H.set_conv_pieces(1)
    try:
        y = H.conv2d_ring(x.to(DEV), wt.to(DEV), b.to(DEV), aff=aff.to(DEV), prologue=pro).cpu()
    finally:
        H.set_conv_pieces(2)
    e = rel_rms(y, truth)
    print(f"fp16 mode, prologue {pro}: rel rms vs fp64 {e:.2e}")
    assert 3e-5 < e < 6e-4


def test_attention_one_product(H):
    B, C, N, heads = 2, 512, 1024, 8
    qkv = rnd(21, B, 3 * C, N)
    d = C // heads
    q, k, v = (t.double().reshape(B, heads, d, N) for t in qkv.split(C, 1))
    p = torch.softmax(torch.einsum("bhdn,bhdm->bhnm", q, k) / math.sqrt(d), -1)
    truth = torch.einsum("bhnm,bhdm->bhdn", p, v).reshape(B, C, N)
    H.set_conv_pieces(1)
    try:
        y = H.attention(qkv.to(DEV), heads).cpu()
    finally:
        H.set_conv_pieces(2)
    y2 = H.attention(qkv.to(DEV), heads).cpu()
    e1, e2 = rel_rms(y, truth), rel_rms(y2, truth)
    print(f"attention: fp16 mode rel rms {e1:.2e}; parity mode {e2:.2e}")
    assert e2 < 1e-6 and 1e-5 < e1 < 3e-3


def test_unet_and_sampler_tolerance_class(O):
    """The stated tolerance class of the mode, measured at 64x1024 (synthetic untrained network, the hardest case: its gains are
    higher than a trained one's):
      U-Net forward vs the fp64 oracle      rel rms < 3e-3, measured 1.5e-3   (parity mode: 1.3e-6; an fp16-autocast torch forward: same class)
      48-step DDPM final sample vs the parity mode on the same noise, [-1, 1] range image: rms < 1e-3, 99th percentile
      < 4e-3, max < 5e-2 (measured 2.4e-4 / 9.9e-4 / 7.1e-3; the parity mode itself: max 7e-6 against the fp64 oracle).
    Switching back restores the parity mode bit for bit; the mode is never the default."""
    import r2dm_amd

    ck = synthetic_ckpt()
    ddpm, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=2)
    assert ddpm.model.precision == "fp32"
    x, c = rnd(70, 2, 2, 64, 1024).to(DEV), torch.tensor([-3.0, 2.0], device=DEV)
    sd = {k: v.double().to(DEV) for k, v in O.strip_prefix(ck["ema_weights"]).items()}
    truth = O.unet_forward(sd, O.UNetConfig(), x.double(), c.double()).cpu()
    y32 = ddpm.model(x, c).cpu()
    ddpm.model.set_precision("fp16")
    y16 = ddpm.model(x, c).cpu()
    e32, e16 = rel_rms(y32, truth), rel_rms(y16, truth)
    print(f"U-Net 64x1024 vs fp64: fp16 mode rel rms {e16:.2e} max {max_abs(y16, truth):.2e} | parity mode rel rms {e32:.2e}")
    assert e32 < 3e-6 and 2e-4 < e16 < 3e-3

    rng = lambda: r2dm_amd.setup_rng([0, 1], DEV)
    s16 = ddpm.sample(batch_size=2, num_steps=48, progress=False, rng=rng()).clamp(-1, 1)
    ddpm.model.set_precision("fp32")
    s32 = ddpm.sample(batch_size=2, num_steps=48, progress=False, rng=rng()).clamp(-1, 1)
    assert torch.equal(ddpm.model(x, c).cpu(), y32)
    d = (s16 - s32).abs().flatten().double()
    rms, q99, mx = d.pow(2).mean().sqrt().item(), torch.quantile(d[:: 4], 0.99).item(), d.max().item()
    print(f"48-step DDPM sample, fp16 mode vs parity mode: rms {rms:.2e} q99 {q99:.2e} max {mx:.2e}")
    assert torch.isfinite(s16).all() and rms < 1e-3 and q99 < 4e-3 and mx < 5e-2  # measured 2.4e-4 / 9.9e-4 / 7.1e-3


def test_mode_plumbing(tmp_path):
    """setup_model(precision="fp16"), the deprecated alias, sample_and_save.py --precision fp16."""
    import subprocess
    import sys

    import r2dm_amd

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ck = synthetic_ckpt(resolution=(64, 1024))
    ddpm, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=1, precision="fp16")
    assert ddpm.model.precision == "fp16"
    a = ddpm.sample(batch_size=1, num_steps=2, progress=False, rng=r2dm_amd.setup_rng([3], DEV))
    path = tmp_path / "ck.pth"
    torch.save(ck, path)
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "sample_and_save.py"), "--ckpt", str(path), "--output_dir", str(out), "--batch_size", "1",
                        "--num_samples", "1", "--num_steps", "2", "--precision", "fp16"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert (out / "samples_0000000000.pth").exists() and torch.isfinite(a).all()
