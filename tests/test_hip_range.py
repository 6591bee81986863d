"""-m gpu: the default (22-bit split fp16 operand) kernels OFF the happy path -- VERDICT round 2, weak #2 / #3.

Every other kernel test draws O(1) Gaussian operands.  Here: wide dynamic range, globally tiny weights (what the reference's
zero-initialised tensors become after training: /root/reference/models/ops.py:9-11, efficient_unet.py:39,84,267), tiny
activations (the split's absolute floor), large GroupNorm / AdaGN gains (the range guard must follow the data, not a
worst-case estimate), and the guard's coverage where the fp32-MFMA kernel produces the operands of an fp16 consumer.

The arithmetic under test: v = h + 2^-11 l with h = RNE_f16(v), l = RNE_f16(2^11 (v - h)); x w = xh wh + 2^-11 (xh wl + xl wh),
the xl wl term (2^-22 relative) dropped.  Weights are scaled per layer by a power of two so that max|w| is in [2^9, 2^10)
(exact; undone in the epilogue).  The representation has an ABSOLUTE floor: below |v| ~ 6e-5 h is an fp16 subnormal and below
|v - h| ~ 3e-8 so is l, whose quantum 2^-24 then bounds the error of v at 2^-36 ~ 1.5e-11 -- relative to the layer's largest
weight (after scaling: 2^-46) or, for activations, in absolute terms."""
import math
import os

import pytest
import torch

from conftest import GOLDEN_RES, max_abs, rnd, synthetic_ckpt

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
    return ((a.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()


def log_uniform(seed, lo, hi, *shape):
    """sign * 10^U(lo, hi)"""
    g = torch.Generator().manual_seed(seed)
    mag = 10.0 ** (torch.rand(*shape, generator=g, dtype=torch.float64) * (hi - lo) + lo)
    sign = torch.where(torch.rand(*shape, generator=g) < 0.5, -1.0, 1.0).double()
    return (mag * sign).float()


SHAPES = [(3, 64, 64, 16, 256), (3, 256, 256, 16, 256), (1, 256, 768, 8, 128), (1, 512, 512, 8, 128)]  # (ksize, cin, cout, h, w)


@pytest.mark.parametrize("k,cin,cout,h,w", SHAPES)
@pytest.mark.parametrize("wscale", [1.0, 1e-4, 1e-6, 1e-9])
def test_tiny_weights_keep_fp32_class_relative_accuracy(O, H, k, cin, cout, h, w, wscale):
    """A layer whose weights are globally tiny (a trained zero-initialised conv2 / out_proj): the relative error of the
    split-operand kernels against fp64 must not depend on the weights' scale -- as it does not for an fp32 FMA chain.
    Without the per-layer power-of-two scale of the packers the error at 1e-6 was ~1e-5 relative (l in the fp16 subnormals)."""
    x = rnd(1, 2, cin, h, w)
    wt = rnd(2, cout, cin, k, k) / math.sqrt(k * k * cin) * wscale
    b = torch.zeros(cout)
    ref = O.conv_ring(x.double(), wt.double(), b.double())
    y = H.conv2d_ring(x.to(DEV), wt.to(DEV), b.to(DEV)).cpu()
    f32 = O.conv_ring(x, wt, b)  # the reference's own fp32 arithmetic (torch CPU) as the yardstick
    e_hip, e_f32 = rel_rms(y, ref), rel_rms(f32, ref)
    print(f"k={k} {cin}->{cout} weights x{wscale:g}: rel rms hip {e_hip:.2e}, fp32 cpu {e_f32:.2e}")
    assert e_hip < 4e-7 and e_hip < 2.0 * e_f32 + 5e-8


@pytest.mark.parametrize("k,cin,cout,h,w", SHAPES[:3])
def test_log_uniform_operands(O, H, k, cin, cout, h, w):
    """Operand magnitudes log-uniform over eleven decades (1e-8 ... 1e3; weights 1e-8 ... 1, then 1/sqrt(K)): the output is
    dominated by the largest products, and the error relative to the output's scale must be fp32-class (no cliff from the
    small operands' subnormal pieces, no overflow of the large ones)."""
    x = log_uniform(5, -8, 3, 2, cin, h, w)
    wt = log_uniform(6, -8, 0, cout, cin, k, k) / math.sqrt(k * k * cin)
    b = rnd(7, cout)
    ref = O.conv_ring(x.double(), wt.double(), b.double())
    y = H.conv2d_ring(x.to(DEV), wt.to(DEV), b.to(DEV)).cpu()
    f32 = O.conv_ring(x, wt, b)
    e_hip, e_f32 = rel_rms(y, ref), rel_rms(f32, ref)
    print(f"k={k} {cin}->{cout} log-uniform: rel rms hip {e_hip:.2e}, fp32 cpu {e_f32:.2e}; max|y| {ref.abs().max():.1f}")
    assert torch.isfinite(y).all() and e_hip < 4e-7 and e_hip < 2.0 * e_f32 + 5e-8


@pytest.mark.parametrize("xscale", [1e-3, 1e-6])
def test_tiny_activations_absolute_floor(O, H, xscale):
    """Raw (not GroupNorm-normalised) activations that are globally tiny sit on the split's absolute floor: every element is
    represented to ~1.5e-11 (2^-36), whatever its size.  With K = 9 x 64 products of |w| ~ 1/24 that is ~1.5e-11 on the
    output -- fp32 keeps 6e-8 relative instead, so at |x| ~ 1e-6 the split is ~1e-5 relative: stated here and in DESIGN.md,
    asserted as an absolute bound.  (GroupNorm inputs are renormalised and never get here; the network's raw inputs --
    residual stream, attention output -- are O(1).)"""
    cin, cout, h, w = 64, 64, 16, 256
    x, wt, b = rnd(1, 2, cin, h, w) * xscale, rnd(2, cout, cin, 3, 3) / math.sqrt(9 * cin), torch.zeros(cout)
    ref = O.conv_ring(x.double(), wt.double(), b.double())
    y = H.conv2d_ring(x.to(DEV), wt.to(DEV), b.to(DEV)).cpu()
    err = (y.double() - ref).pow(2).mean().sqrt().item()
    print(f"activations x{xscale:g}: abs rms error {err:.2e} (output rms {ref.pow(2).mean().sqrt():.2e})")
    assert err < max(3e-11, 3e-7 * ref.pow(2).mean().sqrt().item())


@pytest.mark.parametrize("B,C,N", [(2, 256, 1024)])
def test_attention_wide_dynamic_range(O, H, B, C, N):
    """q, k small (flat softmax) and v log-uniform over eight decades: the fp16-pipe attention core against fp64."""
    qk = rnd(11, B, 2 * C, N) * 0.05
    v = log_uniform(12, -6, 2, B, C, N)
    qkv = torch.cat([qk, v], 1).contiguous()
    ref = O.attention_core(qkv.double(), 8) if hasattr(O, "attention_core") else None
    if ref is None:
        d = C // 8
        q, kk, vv = (t.double().reshape(B, 8, d, N) for t in qkv.split(C, 1))
        p = torch.softmax(torch.einsum("bhdn,bhdm->bhnm", q, kk) / math.sqrt(d), -1)
        ref = torch.einsum("bhnm,bhdm->bhdn", p, vv).reshape(B, C, N)
    y = H.attention(qkv.to(DEV), 8).cpu()
    e = rel_rms(y, ref)
    print(f"attention wide range: rel rms {e:.2e}")
    assert torch.isfinite(y).all() and e < 5e-7


# ---- the range guard follows the data -----------------------------------------------------------------------------------
def _edited(ckpt, edit):
    ck = dict(ckpt)
    ck["ema_weights"] = dict(ck["ema_weights"])
    edit(ck["ema_weights"])
    return ck


def test_large_adagn_scales_and_groupnorm_gains_run_in_the_default_mode(O):
    """AdaGN scales ~ N(0, 3) (through the projection biases) and GroupNorm gains of +-100 on every channel -- far beyond
    |gamma'| ~ 90, where the round-2 worst-case (Samuelson) bound refused the forward -- are ordinary numbers for the data:
    |gamma' x_hat| stays below ~1e3.  The default mode must run them, agree with the fp64 oracle, and leave nothing pending.
    (The guard is |a| M + |d| with M = the square root of the largest statistics-slot energy >= max|x|: 22-45 sigma for slots
    of 512-2048 elements, so it trips from |gamma'| ~ 1.4e3 -- not 90 -- and costs the producing kernels nothing; recording
    the exact maximum in the convolution epilogues was built, measured at +1.1 % of the step, and dropped.)"""
    import r2dm_amd

    g = torch.Generator().manual_seed(3)

    def edit(w):
        for k in list(w):
            if k.endswith("norm2.proj.1.bias"):
                C = w[k].numel() // 2
                nb = w[k].clone()
                nb[:C] = torch.randn(C, generator=g) * 3.0
                w[k] = nb
            if k.endswith("norm1.weight"):
                w[k] = torch.where(torch.rand(w[k].shape, generator=g) < 0.5, -100.0, 100.0)

    ck = _edited(synthetic_ckpt(), edit)
    ddpm, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=1)
    x, c = rnd(95, 1, 2, 64, 1024).to(DEV), torch.tensor([0.5], device=DEV)
    y = ddpm.model(x, c)  # (raises R2DMError if the guard trips)
    sd = {k: v.double().to(DEV) for k, v in O.strip_prefix(ck["ema_weights"]).items()}
    truth = O.unet_forward(sd, O.UNetConfig(), x.double(), c.double())
    e = rel_rms(y.cpu(), truth.cpu())
    print(f"gains +-100, AdaGN scales N(0,3): rel rms vs fp64 {e:.2e}, output rms {truth.pow(2).mean().sqrt().item():.3g}")
    assert torch.isfinite(y).all() and e < 5e-6
    ddpm.model.check_range()  # nothing pending


def test_gains_that_really_leave_the_fp16_range_fail_loudly():
    """A GroupNorm gain of 3e4 puts |gamma x_hat| ~ 1e5 > 65504 on the data themselves: the forward must fail (not return
    saturated numbers), immediately and -- inside a sampling loop -- after the first step rather than after the last; the
    bf16x3 mode (fp32 operand range) runs the same weights."""
    import r2dm_amd
    from r2dm_amd._lib import R2DMError

    def edit(w):
        k = next(k for k in w if k.endswith("d_block1.residual_blocks.0.norm1.weight"))
        w[k] = torch.full_like(w[k], 3.0e4)

    ddpm, _, _ = r2dm_amd.setup_model(_edited(synthetic_ckpt(), edit), device=DEV, show_info=False, max_batch=2, strict_range=True)
    x, c = rnd(95, 2, 2, 64, 1024).to(DEV), torch.zeros(2, device=DEV)
    with pytest.raises(R2DMError, match="fp16 range"):
        ddpm.model(x, c)
    calls = []
    fwd = ddpm.model.forward
    ddpm.model.forward = lambda *a, **k: (calls.append(1), fwd(*a, **k))[1]
    with pytest.raises(R2DMError, match="fp16 range"):
        ddpm.sample(batch_size=2, num_steps=12, progress=False)
    assert len(calls) == 1  # the early check: one denoiser call, not twelve
    ddpm.model.forward = fwd
    ddpm.model.set_precision("fp32-bf16x3")
    assert torch.isfinite(ddpm.model(x, c)).all()
    ddpm.model.check_range()


def test_tripped_range_guard_falls_back_to_the_wide_range_split():
    """VERDICT round 3, missing #3: the reference samples any finite checkpoint (models/efficient_unet.py:269-295); the drop-in
    raised after step 1 and asked the user to re-run.  Default now (strict_range=False): the 3e4-gain checkpoint of the test above
    gives ONE RuntimeWarning, the model switches itself to 'fp32-bf16x3', the call is repeated -- a forward and a seeded 12-step
    sample come out bit-identical to a model that was set up with precision='fp32-bf16x3' in the first place, with one extra
    denoiser call in the loop."""
    import warnings

    import r2dm_amd

    def edit(w):
        k = next(k for k in w if k.endswith("d_block1.residual_blocks.0.norm1.weight"))
        w[k] = torch.full_like(w[k], 3.0e4)

    ck = _edited(synthetic_ckpt(), edit)
    wide, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=2, precision="fp32-bf16x3")
    x, c = rnd(95, 2, 2, 64, 1024).to(DEV), torch.zeros(2, device=DEV)
    y_wide = wide.model(x, c)
    s_wide = wide.sample(batch_size=2, num_steps=12, progress=False, rng=r2dm_amd.setup_rng([5, 6], DEV))
    # (a) a stand-alone forward
    a, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=2)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        y = a.model(x, c)
        y2 = a.model(x, c)  # (already switched: no second warning)
    assert sum(issubclass(i.category, RuntimeWarning) and "fp32-bf16x3" in str(i.message) for i in w) == 1
    assert a.model.precision == "fp32-bf16x3" and a.model.range_fallbacks == 1
    assert torch.equal(y, y_wide) and torch.equal(y2, y_wide)
    # (b) inside a sampling loop: the first step is repeated, the draws are the same
    b, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=2)
    calls = []
    fwd = b.model.forward
    b.model.forward = lambda *a_, **k: (calls.append(1), fwd(*a_, **k))[1]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        s = b.sample(batch_size=2, num_steps=12, progress=False, rng=r2dm_amd.setup_rng([5, 6], DEV))
    assert sum(issubclass(i.category, RuntimeWarning) for i in w) == 1 and len(calls) == 13
    assert torch.equal(s, s_wide)
    # (c) a short loop too (<= 8 steps: the early check runs whenever the guard is not strict)
    c3, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=2)
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        s3 = c3.sample(batch_size=2, num_steps=3, progress=False, rng=r2dm_amd.setup_rng([5, 6], DEV))
    assert torch.equal(s3, wide.sample(batch_size=2, num_steps=3, progress=False, rng=r2dm_amd.setup_rng([5, 6], DEV)))


def test_attention_operands_are_guarded_where_the_fp32_mfma_kernel_produces_them():
    """at resolutions where proj_f16x2 does not apply (16x128: W/8 = 16 is not a multiple of 64) the
    qkv projection runs on the fp32-MFMA kernel, which did not record max|qkv| -- the fp16 attention core ran unguarded.
    A qkv bias of 1e5 must now make the forward fail; the bf16x3 mode runs it."""
    import r2dm_amd
    from r2dm_amd._lib import R2DMError

    def edit(w):
        k = next(k for k in w if k.endswith("d_block4.self_attn_block.attn.in_proj_bias"))
        w[k] = torch.full_like(w[k], 1.0e5)

    ddpm, _, _ = r2dm_amd.setup_model(_edited(synthetic_ckpt(resolution=GOLDEN_RES), edit), device=DEV, show_info=False, max_batch=2, strict_range=True)
    x, c = rnd(98, 2, 2, *GOLDEN_RES).to(DEV), torch.zeros(2, device=DEV)
    with pytest.raises(R2DMError, match="fp16 range"):
        ddpm.model(x, c)
    ddpm.model.set_precision("fp32-bf16x3")
    assert torch.isfinite(ddpm.model(x, c)).all()


def test_weight_flag_survives_blob_adoption():
    """r2dm_bind_blob cleared the packer's 'weight not usable' flag, so a rank that ADOPTED a broadcast blob
    ran silently where rank 0 raised.  A non-finite convolution weight must fail on the packing model and on the adopter."""
    import r2dm_amd
    from r2dm_amd._lib import R2DMError

    def edit(w):
        k = next(k for k in w if k.endswith("d_block2.residual_blocks.1.conv1.weight"))
        nw = w[k].clone()
        nw[0, 0, 0, 0] = float("inf")
        w[k] = nw

    ck = _edited(synthetic_ckpt(), edit)  # (full size: the layer must be one the fp16 packings cover)
    a, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=2)
    x, c = rnd(99, 2, 2, 64, 1024).to(DEV), torch.zeros(2, device=DEV)
    with pytest.raises(R2DMError, match="not finite"):
        a.model(x, c)
    blob = a.model.packed_weights(DEV).clone()
    b, _, _ = r2dm_amd.setup_model(synthetic_ckpt(), device="cpu", show_info=False, max_batch=2)
    b.to(DEV)
    b.model.adopt_packed_weights(blob)
    with pytest.raises(R2DMError, match="not finite"):
        b.model(x, c)


def test_late_range_guard_trip_is_replayed_from_the_last_checked_step():
    """VERDICT round 4, missing #2: a guard that trips only late in a loop (low t) used to raise after the last step and lose the call.
    A 40-step seeded loop whose guard is raised behind denoiser call 35 (the C ABI's test hook writes the bound a saturating operand
    would have left): the checks after steps 0 and 31 pass, the one at the end trips -- ONE RuntimeWarning, the model switches to
    'fp32-bf16x3', steps 32..39 run again on the recorded noise (8 extra denoiser calls, no new draws), no exception; the sample
    agrees with a model that ran 'fp32-bf16x3' from the start to the parity class of the two splits (steps 0..31 ran on the fp16
    split), and a second call runs clean on the wide split, bit-identical to that model."""
    import warnings

    import r2dm_amd
    from r2dm_amd import _lib

    ck = synthetic_ckpt()
    wide, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=2, precision="fp32-bf16x3")
    s_wide = wide.sample(batch_size=2, num_steps=40, progress=False, rng=r2dm_amd.setup_rng([5, 6], DEV))
    b, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=2)
    calls = []
    fwd = b.model.forward

    def forward(*a, **k):
        y = fwd(*a, **k)
        calls.append(1)
        if len(calls) == 35:
            _lib.check(_lib.lib().r2dm_test_raise_range_bound(b.model._engine.h, 1.0e5, _lib.stream_ptr(y.device)))
        return y

    b.model.forward = forward
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        s = b.sample(batch_size=2, num_steps=40, progress=False, rng=r2dm_amd.setup_rng([5, 6], DEV))
    assert sum(issubclass(i.category, RuntimeWarning) and "fp32-bf16x3" in str(i.message) for i in w) == 1
    assert len(calls) == 48 and b.model.precision == "fp32-bf16x3" and b.model.range_fallbacks == 1
    assert torch.isfinite(s).all() and max_abs(s.cpu(), s_wide.cpu()) < 2e-5
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        s2 = b.sample(batch_size=2, num_steps=40, progress=False, rng=r2dm_amd.setup_rng([5, 6], DEV))
    assert not w and torch.equal(s2, s_wide)
    # return_all keeps one entry per step through a replay
    c, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=2)
    n = []
    fwd_c = c.model.forward
    c.model.forward = lambda *a, **k: (n.append(1), fwd_c(*a, **k), len(n) == 3 and _lib.check(_lib.lib().r2dm_test_raise_range_bound(c.model._engine.h, 1.0e5, _lib.stream_ptr(DEV))))[1]
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        allx = c.sample(batch_size=2, num_steps=6, progress=False, rng=r2dm_amd.setup_rng([5, 6], DEV), return_all=True)
    assert allx.shape[0] == 7 and len(n) == 12
    assert torch.equal(allx[-1], wide.sample(batch_size=2, num_steps=6, progress=False, rng=r2dm_amd.setup_rng([5, 6], DEV)))


def test_checkpoint_preflight_on_both_sides_of_the_trip_point(capsys):
    """VERDICT round 5, item 5 / missing #2: `python -m r2dm_amd.check <ckpt>` -- the sampler's own loop with the guard read after every
    step, per guarded layer the largest bound / 65504 and where it peaked, exit code 1 iff a step would fall back.  Driven over the
    synthetic checkpoint with HEAVY-TAILED AdaGN gains (Cauchy-distributed scales through the projection biases,
    /root/reference/models/ops.py:190-200) scaled to either side of the trip point."""
    from r2dm_amd import check

    def ckpt(c):
        g = torch.Generator().manual_seed(11)

        def edit(w):
            for k in list(w):
                if k.endswith("norm2.proj.1.bias"):
                    C = w[k].numel() // 2
                    nb = w[k].clone()
                    u = torch.rand(C, generator=g)
                    nb[:C] = c * torch.tan(3.14159265 * (u - 0.5)).clamp(-200, 200)  # Cauchy, tails cut at +-200 c
                    w[k] = nb
        return _edited(synthetic_ckpt(), edit)

    # c = 0.3: |1 + scale| up to ~60 -> bounds of a few thousand at most; c = 30: |1 + scale| up to 6000 x (M ~ 20 sigma) >> 65504
    lo = check.preflight(ckpt(0.3), num_steps=8, batch=2, device=DEV)
    assert lo["trips"] == [] and len(lo["sites"]) > 40
    name, b, i, cond = lo["sites"][0]
    assert 100.0 < b < 65504.0 and ".conv2: GroupNorm output bound" in name, lo["sites"][:3]
    hi = check.preflight(ckpt(30.0), num_steps=8, batch=2, device=DEV)
    assert len(hi["trips"]) >= 1 and hi["sites"][0][1] >= 65504.0
    assert all(".conv2: GroupNorm output bound" in t[3] for t in hi["trips"]), hi["trips"][:3]
    # ... and the command line: exit codes 0 / 1, the table names the layer
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        for c, rc in ((0.3, 0), (30.0, 1)):
            path = os.path.join(d, f"ck{rc}.pth")
            torch.save(ckpt(c), path)
            assert check.main([path, "--steps", "4", "--batch", "1", "--device", DEV, "--top", "5"]) == rc
            out = capsys.readouterr().out
            assert "bound / 65504" in out and ("would fall back" in out) == bool(rc), out[-600:]
    # the default synthetic checkpoint itself: far inside the range
    base = check.preflight(synthetic_ckpt(), num_steps=4, batch=1, device=DEV)
    assert base["trips"] == [] and base["sites"][0][1] < 0.05 * 65504.0
    # ... and a discrete-time checkpoint (the sampler walks integer timesteps: /root/reference/models/diffusion/discrete_time.py:182-201)
    disc = check.preflight(synthetic_ckpt(timestep_type="discrete", num_training_steps=1000, noise_schedule="linear"), num_steps=4, batch=1, device=DEV)
    assert disc["condition"] == "timestep" and disc["trips"] == [] and len(disc["sites"]) > 40
