"""-m gpu: the whole denoiser and the samplers through the drop-in API, against the reference's golden
vectors (16x128) and against the oracle at the full 64x1024 size.

Tolerance policy (stated, fp32 everywhere):
  * U-Net forward: max |delta eps| < 2e-5 (measured ~3e-6; the reference's own fp32-vs-fp64 gap is
    ~3e-6).
  * sampling: per-pixel |delta x| < 1e-4 per step / end-to-end -- BASELINE.json's target, which is the
    reference's own fp32 noise floor (x_0 = (x_t - sigma eps)/alpha_t amplifies eps error ~1800x at t~1
    before the clamp).
"""
import os

import pytest
import torch

from conftest import GOLDEN_RES, max_abs, rnd, synthetic_ckpt

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(**kw):
    import r2dm_amd

    ddpm, lidar, cfg = r2dm_amd.setup_model(synthetic_ckpt(**kw), device=DEV, show_info=False)
    return ddpm, lidar


class Tape:
    """Feeds recorded noise draws to ddpm.randn (the reference API has no explicit-noise argument)."""

    def __init__(self, ddpm, noise):
        self.noise, self.i = noise, 0
        ddpm.randn = self

    def __call__(self, *shape, rng=None, **kw):
        z = self.noise[self.i].to(DEV)
        self.i += 1
        assert tuple(z.shape) == tuple(shape)
        return z


@pytest.fixture(scope="module")
def small():
    return build(resolution=GOLDEN_RES)


def test_unet_golden(golden, small):
    g = golden("unet")
    net = small[0].model
    x = g["x"].to(DEV)
    for i, c in enumerate(g["conds"].tolist()):
        y = net(x, torch.full((2,), c, device=DEV)).cpu()
        assert max_abs(y, g["y"][i]) < 2e-5, c
    assert max_abs(net(x, g["cond_mixed"].to(DEV)).cpu(), g["y_mixed"]) < 2e-5
    # scalar timestep broadcast (efficient_unet.py:273-274) and int64 conditions
    assert max_abs(net(x, torch.tensor(0.0, device=DEV)).cpu(), g["y"][2]) < 2e-5


def test_unet_batch_sizes_agree(small):
    """Samples are independent: any batch split gives bit-identical rows (data-parallel invariant)."""
    net = small[0].model
    x, c = rnd(60, 5, 2, *GOLDEN_RES).to(DEV), torch.tensor([-9.0, -1.0, 0.5, 4.0, 12.0], device=DEV)
    full = net(x, c)
    parts = torch.cat([net(x[:2], c[:2]), net(x[2:3], c[2:3]), net(x[3:], c[3:])])
    assert torch.equal(full, parts)
    assert torch.equal(full, net(x, c))  # deterministic


def test_unet_full_size_vs_oracle():
    """64x1024 (BASELINE config): HIP vs the fp64 oracle on the GPU, and the fp32 CPU oracle for scale."""
    from oracle import r2dm_oracle as O

    ddpm, _ = build()
    ck = synthetic_ckpt()
    sd = O.strip_prefix(ck["ema_weights"])
    cfg = O.UNetConfig()
    x = rnd(61, 1, 2, 64, 1024)
    errs = {}
    for c in (-15.0, 0.0, 6.0):
        cond = torch.full((1,), c)
        y = ddpm.model(x.to(DEV), cond.to(DEV)).cpu()
        sd64 = {k: v.double().to(DEV) for k, v in sd.items()}
        ref64 = O.unet_forward(sd64, cfg, x.double().to(DEV), cond.double().to(DEV)).cpu()
        errs[c] = max_abs(y, ref64)
        assert errs[c] < 8e-6, errs  # measured 2.0-2.7e-6 (round 2 default path); the fp32 CPU oracle: 1.5e-6
    ref32 = O.unet_forward(sd, cfg, x, torch.full((1,), 6.0))
    assert max_abs(ref32, ref64) < 6e-6  # the oracle's own fp32 error is of the same class (measured 1.5e-6)
    print("unet 64x1024 max|hip - fp64 oracle|:", errs, " fp32 oracle vs fp64:", max_abs(ref32, ref64))


@pytest.mark.parametrize("obj", ["eps", "v", "x_0"])
def test_p_step_golden(golden, obj):
    g = golden("p_step")
    ddpm, _ = build(resolution=GOLDEN_RES, prediction_type=obj)
    x_t = g["x_t"].to(DEV)
    for mode, eta in (("ddpm", 0.0), ("ddim", 0.0), ("ddim", 0.5)):
        for (t, s) in ((1.0, 0.875), (0.5, 0.375), (0.125, 0.0)):
            ddpm.randn = lambda *shape, rng=None, **kw: g["z"].to(DEV)
            y = ddpm.p_step(x_t, torch.full((2,), t), torch.full((2,), s), rng=None, mode=mode, ddim_eta=eta).cpu()
            want = g[f"{obj}_{mode}_{eta}_{t}_{s}"]
            # t = 1 with the eps objective divides by alpha_t = 5.5e-4: the few pixels that escape the +-1 clamp carry
            # a ~1800x copy of the fp32-roundoff difference between two U-Net evaluations (see test_sample_golden);
            # everywhere else, and for all other (t, objective) pairs, 1e-4 holds with a wide margin.
            tol = 3e-4 if (t == 1.0 and obj == "eps") else 1e-4
            assert max_abs(y, want) < tol, (mode, eta, t, s)
            assert torch.quantile((y - want).abs().flatten(), 0.99).item() < 2e-6, (mode, eta, t, s)


def _fp64_truth(noise, mode, S, res=GOLDEN_RES):
    """fp64 oracle on the CPU driven by the same noise tape and by the same float32 schedule scalars the
    reference uses ((B,)-shaped evaluation on the host): 'exact arithmetic' of the same algorithm."""
    from oracle import r2dm_oracle as O

    sd = {k: v.double() for k, v in O.strip_prefix(synthetic_ckpt(resolution=res)["ema_weights"]).items()}
    cfg = O.UNetConfig(resolution=res)
    return O.sample_continuous(lambda x, c: O.unet_forward(sd, cfg, x, c), tuple(noise[0].shape), S, noises=list(noise),
                               return_all=True, mode=mode, device="cpu", dtype=torch.float64)


def rms(a, b):
    return (a.double() - b.double()).pow(2).mean().sqrt().item()


def q99(a, b):
    """99th percentile of |a - b|: blind to the handful of pixels that escape the clamp at t ~ 1 (see below)."""
    return torch.quantile((a.double() - b.double()).abs().flatten(), 0.99).item()


@pytest.mark.parametrize("mode", ["ddpm", "ddim"])
def test_sample_golden(golden, mode):
    """End-to-end sampler vs the reference's own run (golden) on the same noise tape.

    Both are fp32 evaluations of an ill-conditioned map: at t ~ 1, x_0 = (x_t - sigma*eps)/alpha_t with
    1/alpha_t ~ 1800, so the handful of pixels that escape the +-1 clamp carry a ~350x amplified copy of the
    ~1e-7..1e-6 difference between two fp32 U-Net evaluations, while clamped pixels agree exactly.  Max-norm
    differences at the first steps are therefore a lottery over a few pixels (the reference itself moves by
    1e-5..1.4e-4 against fp64 arithmetic depending on the noise draw), and at this 8192-pixel size even the RMS of
    step 1 is set by 2-3 such pixels: two HIP builds with the SAME U-Net accuracy (rms 2.7e-7 vs 2.8e-7 against fp64,
    scripts/diag_unet.py) gave 0.9e-6 and 2.1e-6 here.  The robust statements are
      * the 99th percentile of |error vs fp64 truth| within 2.5x of the reference's (measured 0.6x .. 1.8x; the U-Net's
        convolutions carry ~1.7x the accumulation roundoff of oneDNN's blocked fp32 sums -- per-layer budget in
        profiles/r02a_error_budget_default.txt; the fused SiLU approximations are NOT the cause: ablation there),
        and RMS(hip - reference) < 1e-5, RMS vs truth < 3.5e-6 at every step (measured <= 1.6e-6);
      * max|hip - reference| < 3e-4 (DDPM; errors contract) / 1.5e-3 (DDIM: no fresh noise, errors persist);
      * the final DDPM sample -- what BASELINE.json's "per-pixel delta < 1e-4" is about -- within 1e-4."""
    g = golden(f"sample_{mode}")
    ddpm, _ = build(resolution=GOLDEN_RES)
    Tape(ddpm, g["noise"])
    out = ddpm.sample(batch_size=2, num_steps=8, progress=False, rng=None, return_all=True, mode=mode).cpu()
    assert out.shape == g["out"].shape
    truth = _fp64_truth(g["noise"], mode, 8)
    rows = []
    for i in range(out.shape[0]):
        rows.append((max_abs(out[i], g["out"][i]), rms(out[i], g["out"][i]), rms(out[i], truth[i]), rms(g["out"][i], truth[i]),
                     q99(out[i], truth[i]), q99(g["out"][i], truth[i])))
    print(f"sample_golden[{mode}] per step (max hip-ref, rms hip-ref, rms hip-fp64, rms ref-fp64, q99 hip-fp64, q99 ref-fp64):")
    for r in rows:
        print("   " + "  ".join(f"{v:.2e}" for v in r))
    for i, (mx, r_hr, r_ht, r_rt, q_ht, q_rt) in enumerate(rows):
        assert r_hr < (1e-5 if mode == "ddpm" else 2e-5), (i, rows[i])  # (DDIM carries the step-1 lottery to the end)
        assert q_ht <= max(2.5 * q_rt, 1e-6), (i, rows[i])
        assert r_ht <= (3.5e-6 if mode == "ddpm" else 2e-5), (i, rows[i])  # (DDIM: no fresh noise, errors persist; measured 8.4e-6)
        assert mx < (3e-4 if mode == "ddpm" else 1.5e-3), (i, rows[i])
    if mode == "ddpm":
        assert rows[-1][0] < 1e-4
    Tape(ddpm, g["noise"])
    last = ddpm.sample(batch_size=2, num_steps=8, progress=False, rng=None, return_all=False, mode=mode).cpu()
    assert torch.equal(last, out[-1])


def test_discrete_golden(golden):
    g = golden("discrete")
    ddpm, _ = build(resolution=GOLDEN_RES, timestep_type="discrete", num_training_steps=1000, noise_schedule="linear")
    assert torch.equal(ddpm.beta.cpu(), g["beta"]) and torch.equal(ddpm.alpha_bar.cpu(), g["alpha_bar"])
    for mode in ("ddpm", "ddim"):
        for st in (999, 500, 0):
            ddpm.randn = lambda *shape, rng=None, **kw: g["z"].to(DEV)
            y = ddpm.p_step(g["x_t"].to(DEV), torch.full((2,), st).long(), rng=None, mode=mode).cpu()
            assert max_abs(y, g[f"{mode}_{st}"]) < 1e-4, (mode, st)
    Tape(ddpm, g["sample_noise"])
    out = ddpm.sample(batch_size=2, num_steps=16, progress=False, rng=None, return_all=True, mode="ddpm").cpu()
    assert max_abs(out, g["sample_out"]) < 3e-4


def test_per_sample_noise_rows_equal_the_reference_stack():
    """base.py:81-85 stacks one torch.randn per sample generator; GaussianDiffusion.randn draws straight into the rows of the result (no copy
    kernel in the step loop): the same values, generator states advanced alike -- on the device generators the samplers use."""
    import r2dm_amd

    ddpm, _, _ = r2dm_amd.setup_model(synthetic_ckpt(), device=DEV, show_info=False)
    shape = (5, 2, *GOLDEN_RES)
    for _ in range(2):  # (second round: the states left by the first)
        ga, gb = r2dm_amd.setup_rng(list(range(5)), DEV), r2dm_amd.setup_rng(list(range(5)), DEV)
        for _ in range(3):
            want = torch.stack([torch.randn(*shape[1:], generator=r, device=DEV) for r in ga])
            got = ddpm.randn(*shape, rng=gb, device=DEV)
            assert torch.equal(got, want)


def test_seeded_sampling_partition_invariant(small):
    """Per-sample generators (utils/inference.py:113-114): a sample depends on its seed only, not on
    which batch / rank drew it -- the property multi-GPU sharding relies on (sample_and_save.py:37-46,75)."""
    import r2dm_amd

    ddpm = small[0]
    a = ddpm.sample(batch_size=4, num_steps=3, progress=False, rng=r2dm_amd.setup_rng([0, 1, 2, 3], DEV))
    b = ddpm.sample(batch_size=2, num_steps=3, progress=False, rng=r2dm_amd.setup_rng([2, 3], DEV))
    assert torch.equal(a[2:], b)
    assert a.abs().max() <= 8 and torch.isfinite(a).all()


def test_sample_full_size_vs_oracle():
    """64x1024, 4 DDPM steps on GPU-generator noise: HIP sampler vs the fp32 oracle on the CPU (the reference's
    arithmetic) and vs fp64 truth.  (The oracle run through MIOpen on the GPU is NOT used as a yardstick.)"""
    import r2dm_amd
    from oracle import r2dm_oracle as O

    ddpm, _ = build()
    sd = O.strip_prefix(synthetic_ckpt()["ema_weights"])
    cfg = O.UNetConfig()
    rng = r2dm_amd.setup_rng([7], DEV)
    noise = [ddpm.randn(1, 2, 64, 1024, rng=rng, device=DEV) for _ in range(5)]
    Tape(ddpm, noise)
    got = ddpm.sample(batch_size=1, num_steps=4, progress=False, rng=None, return_all=True).cpu()
    cpu_noise = [z.cpu() for z in noise]
    want = O.sample_continuous(lambda x, c: O.unet_forward(sd, cfg, x, c), (1, 2, 64, 1024), 4, noises=cpu_noise, return_all=True)
    truth = _fp64_truth(cpu_noise, "ddpm", 4, res=(64, 1024))
    rows = [(max_abs(got[i], want[i]), rms(got[i], want[i]), rms(got[i], truth[i]), rms(want[i], truth[i]),
             q99(got[i], truth[i]), q99(want[i], truth[i])) for i in range(5)]
    print("sample 64x1024 per step (max hip-cpu32, rms hip-cpu32, rms hip-fp64, rms cpu32-fp64, q99 hip-fp64, q99 cpu32-fp64):")
    for r in rows:
        print("   " + "  ".join(f"{v:.2e}" for v in r))
    # 4 coarse steps of an UNTRAINED (high-gain) network; see test_sample_golden for why the tail pixels are excluded
    # measured: q99 1.0x .. 1.9x of the reference's, rms(hip - fp64) 5.4 .. 6.2e-6 (reference 1.8 .. 2.6e-6), max 1.9e-3
    for mx, r_hr, r_ht, r_rt, q_ht, q_rt in rows:
        assert r_hr < 1.5e-5, rows
        assert q_ht <= max(2.5 * q_rt, 1e-6), rows
        assert r_ht <= max(4 * r_rt, 1.2e-5), rows
        assert mx < 4e-3, rows


def test_north_star_final_sample_parity():
    """BASELINE.json: "outputs match the reference PyTorch sampler on identical noise ... per-pixel output delta
    < 1e-4".  Full-size 64x1024 DDPM, 48 steps (the 256-step run of scripts/validate_256.py gives max 3.7e-6, rms
    2.6e-7; 48 steps keep the CPU oracle at ~30 s): the FINAL sample against the oracle -- the reference's arithmetic,
    torch fp32 on the CPU -- on the same noise tape.  Intermediate steps near t = 1 are ill-conditioned (see
    test_sample_golden); the errors contract as alpha_t grows and the finished sample agrees to a few 1e-6."""
    import r2dm_amd
    from oracle import r2dm_oracle as O

    S = 48
    ddpm, _ = build()
    rng = r2dm_amd.setup_rng([11], DEV)
    tape = [ddpm.randn(1, 2, 64, 1024, rng=rng, device=DEV) for _ in range(S + 1)]
    Tape(ddpm, tape)
    got = ddpm.sample(batch_size=1, num_steps=S, progress=False, rng=None).cpu()
    sd = O.strip_prefix(synthetic_ckpt()["ema_weights"])
    cfg = O.UNetConfig()
    want = O.sample_continuous(lambda x, c: O.unet_forward(sd, cfg, x, c), (1, 2, 64, 1024), S, noises=[z.cpu() for z in tape])
    mx, r = max_abs(got, want), rms(got, want)
    print(f"north star: {S}-step final sample max|hip - oracle| = {mx:.2e}, rms {r:.2e}")
    assert mx < 1e-4 and r < 2e-6


def test_lidar_postprocess_matches_members(golden, small):
    lidar = small[1]
    g = golden("lidar")
    y = lidar.postprocess(g["x"].to(DEV))
    # the fused kernel vs the member functions evaluated by torch on the same GPU (same exp2/sin/cos): tight
    s = lidar.denormalize(g["x"].to(DEV))
    depth = lidar.revert_depth(s[:, [0]])
    want = torch.cat([depth, lidar.to_xyz(depth), s[:, [1]]], 1)
    assert (y[:, :1] > 0).eq(want[:, :1] > 0).all()
    assert max_abs(y, want) < 2e-5


def test_high_res_128x2048_vs_fp64_oracle():
    """BASELINE configs[4] geometry (128x2048: 36 Fourier channels -> in_conv 38 ch, 4096 attention tokens):
    U-Net forward vs the fp64 oracle evaluated with torch ops on the GPU, plus batch-split invariance."""
    from oracle import r2dm_oracle as O

    res = (128, 2048)
    ddpm, _ = build(resolution=res)
    sd = O.strip_prefix(synthetic_ckpt(resolution=res)["ema_weights"])
    assert sd["in_conv.weight"].shape[1] == 38
    cfg = O.UNetConfig(resolution=res)
    x = rnd(70, 2, 2, *res)
    cond = torch.tensor([-4.0, 9.0])
    y = ddpm.model(x.to(DEV), cond.to(DEV))
    sd64 = {k: v.double().to(DEV) for k, v in sd.items()}
    sd64["coords"] = sd["coords"].double().to(DEV)
    ref = O.unet_forward(sd64, cfg, x[:1].double().to(DEV), cond[:1].double().to(DEV)).cpu()
    err = max_abs(y[:1].cpu(), ref)
    print("unet 128x2048 max|hip - fp64 oracle|:", err)
    assert err < 2e-5
    assert torch.equal(y[1:], ddpm.model(x[1:].to(DEV), cond[1:].to(DEV)))


def test_batch32_ddim_runs_and_is_seed_determined():
    """BASELINE configs[2] shape: 64x1024, DDIM, batch 32 (3 steps here): finite, clamped range, and sample i
    equals the same seed drawn in a batch of 8 (engine tiling chosen for max_batch=32 vs 8)."""
    import r2dm_amd

    ck = synthetic_ckpt()
    big, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=32)
    small, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=8)
    a = big.sample(batch_size=32, num_steps=3, progress=False, mode="ddim", rng=r2dm_amd.setup_rng(range(32), DEV))
    b = small.sample(batch_size=8, num_steps=3, progress=False, mode="ddim", rng=r2dm_amd.setup_rng(range(8, 16), DEV))
    assert a.shape == (32, 2, 64, 1024) and torch.isfinite(a).all()
    # different layer tilings (128- vs 64-channel tiles) change the fp32 summation order; 3 coarse DDIM steps amplify
    # that ~1e-6 difference in the few unclamped pixels (see test_sample_golden): the robust statement is the RMS
    assert (a[8:16] - b).pow(2).mean().sqrt() < 5e-5 and max_abs(a[8:16], b) < 5e-2


@pytest.mark.skipif(os.environ.get("R2DM_CONV_ALGO", "").startswith("f"), reason="fp32-MFMA algorithm forced")
def test_operand_split_modes_are_both_parity_modes():
    """`precision="fp32"` (f16x2 split in the residual blocks, default) and `"fp32-bf16x3"` (three bf16 pieces everywhere,
    the round-1 mode) at 64x1024, batch 2, against the fp64 oracle: both within the parity bars of
    test_unet_full_size_vs_oracle; the default mode is the more accurate one (measured rms 1.9e-7 vs 2.6e-7; the CPU fp32
    oracle: 1.5e-7).  Switching modes back and forth reproduces each mode bit for bit."""
    from oracle import r2dm_oracle as O

    ddpm, _ = build()
    net = ddpm.model
    ck = synthetic_ckpt()
    sd = O.strip_prefix(ck["ema_weights"])
    x, c = rnd(70, 2, 2, 64, 1024), torch.tensor([-3.0, 2.0])
    truth = O.unet_forward({k: v.double().to(DEV) for k, v in sd.items()}, O.UNetConfig(), x.double().to(DEV), c.double().to(DEV)).cpu()
    ya = net(x.to(DEV), c.to(DEV)).cpu()
    net.set_precision("fp32-bf16x3")
    yb = net(x.to(DEV), c.to(DEV)).cpu()
    net.set_precision("fp32")
    assert torch.equal(net(x.to(DEV), c.to(DEV)).cpu(), ya) and not torch.equal(ya, yb)
    ra, rb = rms(ya, truth), rms(yb, truth)
    # the yardstick (bars relative to an fp32 evaluation measured in the same test, not to each other): the
    # reference's own arithmetic on this GPU -- the oracle in fp32 on stock PyTorch-ROCm (MIOpen / rocBLAS)
    yr = O.unet_forward({k: v.to(DEV) for k, v in sd.items()}, O.UNetConfig(), x.to(DEV), c.to(DEV)).cpu()
    rr = rms(yr, truth)
    print(f"U-Net 64x1024 vs fp64: f16x2 mode rms {ra:.2e} max {max_abs(ya, truth):.2e} | bf16x3 mode rms {rb:.2e} max {max_abs(yb, truth):.2e} | "
          f"the reference's fp32 on PyTorch-ROCm rms {rr:.2e} max {max_abs(yr, truth):.2e}")
    assert ra < 4e-7 and rb < 5.5e-7 and max_abs(ya, truth) < 8e-6 and max_abs(yb, truth) < 8e-6  # ~2x measured
    assert ra <= rr and rb <= rr  # both parity modes are at least as close to exact arithmetic as the reference's fp32 GPU evaluation
    with pytest.warns(DeprecationWarning):
        net.set_precision("bf16x2")  # round 1's reduced mode: kept as a deprecated alias of the (faster, more accurate) default
    assert net.precision == "fp32" and torch.equal(net(x.to(DEV), c.to(DEV)).cpu(), ya)
    with pytest.raises(ValueError):
        net.set_precision("int8")


def test_repaint_kernels_replay_reference_ops():
    """r2dm_repaint_blend / r2dm_q_step vs the reference expressions evaluated op by op by torch on the same GPU
    (continuous_time.py:175,186-189,296): bit-exact (no FMA contraction in the kernels)."""
    from r2dm_amd import _lib

    known, noise, unknown = (rnd(70 + i, 3, 2, *GOLDEN_RES).to(DEV) for i in range(3))
    coef = torch.tensor([[0.3, 0.95], [0.9991, 0.0421], [1.0, 0.0]], device=DEV)
    a, sg = coef[:, 0].view(3, 1, 1, 1), coef[:, 1].view(3, 1, 1, 1)
    for mask in ((rnd(73, 3, 1, *GOLDEN_RES) > 0).float().to(DEV), (rnd(74, 3, 2, *GOLDEN_RES) > 0.5).float().to(DEV),
                 torch.rand(3, 1, *GOLDEN_RES, device=DEV)):
        want = mask * (known * a + noise * sg) + (1 - mask) * unknown
        assert torch.equal(_lib.repaint_blend(known, noise, unknown, mask, coef), want)
    assert torch.equal(_lib.q_step(known, noise, coef), known * a + sg * noise)


def test_repaint_golden(golden):
    """RePaint end to end vs the reference's own run (golden) on the same noise tape; same statistics as
    test_sample_golden (the reverse sub-steps are p_steps: a few pixels near t = 1 carry amplified roundoff)."""
    g = golden("repaint")
    ddpm, _ = build(resolution=GOLDEN_RES)
    Tape(ddpm, g["noise"])
    out = ddpm.repaint(g["known"].to(DEV), g["mask"].to(DEV), num_steps=3, num_resample_steps=2, jump_length=2,
                       progress=False, rng=None, return_all=True).cpu()
    assert out.shape == g["out"].shape
    rows = [(max_abs(out[i], g["out"][i]), rms(out[i], g["out"][i]), q99(out[i], g["out"][i])) for i in range(out.shape[0])]
    print("repaint_golden per output (max, rms, q99 of |hip - reference|):")
    for r in rows:
        print("   " + "  ".join(f"{v:.2e}" for v in r))
    known_px = g["mask"].expand_as(out[-1]) > 0
    for i, (mx, r, q) in enumerate(rows):
        assert r < 1e-5 and q < 5e-6 and mx < 1e-3, (i, rows[i])
    assert rows[-1][0] < 1e-4  # the finished completion: BASELINE's per-pixel bar
    # where the mask is 1 the last output is the known image re-noised at s = 0, where sigma = 5.5e-4 (logSNR +15)
    assert max_abs(out[-1][known_px], g["known"].expand_as(out[-1])[known_px]) < 4e-3


def test_repaint_keeps_known_region_statistics(small):
    """RePaint (continuous_time.py:260-317) on top of the HIP p_step: with an all-ones mask the result is the
    forward-diffused known image at the last step, i.e. (t -> 0) the known image itself."""
    ddpm = small[0]
    known = rnd(80, 2, 2, *GOLDEN_RES).clamp(-1, 1).to(DEV)
    full = torch.ones_like(known)
    out = ddpm.repaint(known, full, num_steps=4, num_resample_steps=2, progress=False)
    assert max_abs(out, known) < 5e-3  # alpha(0) ~ 1, sigma(0) ~ 5.5e-4
    half = full.clone()
    half[..., GOLDEN_RES[1] // 2:] = 0
    out = ddpm.repaint(known, half, num_steps=4, num_resample_steps=2, progress=False, return_all=True)
    assert out.shape[0] == 1 + 4 * 2 - 1 and torch.isfinite(out).all()
    assert max_abs(out[-1][..., : GOLDEN_RES[1] // 2], known[..., : GOLDEN_RES[1] // 2]) < 5e-3


@pytest.mark.parametrize("res,batch", [((64, 1024), 8), ((64, 1024), 3), ((128, 2048), 2)])
def test_group_norm_folded_into_its_consumer_is_bit_identical(res, batch):
    """gn_finalize folded into the consuming convolution (conv_f16x2.hip: the staging waves of every block reduce the
    producers' statistics slots of the block's sample while the first pixels are on their way).  The fold repeats gn_finalize_kernel's reduction
    order and arithmetic (gn_math.h), so a forward with it (default) and without it (R2DM_GN_FOLD=0: the separate launches) must agree bit for bit
    -- at batch 8 (every block inside one sample: 45 of 50 norms folded), at batch 3 (some launches refuse the fold) and at 128x2048 (level 1 has
    too many slots and keeps the separate launch).  The range guard's bound travels the same way: a gain outside the fp16 range trips it
    (tests/test_hip_range.py run with the fold on)."""
    import r2dm_amd

    ck = synthetic_ckpt(resolution=res)
    x, c = rnd(7, batch, 2, *res).to(DEV), torch.linspace(-4.0, 6.0, batch).to(DEV)
    outs = {}
    saved = os.environ.get("R2DM_GN_FOLD")
    for mode in ("1", "0"):
        os.environ["R2DM_GN_FOLD"] = mode
        try:
            m, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=batch)
            outs[mode] = (m.model(x, c).clone(), m.model(x, c).clone())
            del m
        finally:
            os.environ.pop("R2DM_GN_FOLD", None)
            if saved is not None:
                os.environ["R2DM_GN_FOLD"] = saved
    assert torch.equal(outs["1"][0], outs["1"][1]) and torch.equal(outs["0"][0], outs["0"][1])
    assert torch.equal(outs["1"][0], outs["0"][0])


@pytest.mark.parametrize("res,batch", [((64, 1024), 8), ((64, 1024), 2), ((128, 2048), 2)])
def test_operand_prepass_with_folded_group_norm_is_bit_identical(res, batch):
    """Round 6: the launches on 32-channel output tiles (u_block4 at batch 8) take their input through the operand pre-pass with the GroupNorm folded
    into it (presplit.hip presplit_fold_kernel: gn_finalize_kernel's reduction and gn_math.h's arithmetic, the stagers' own transform) instead of
    transforming it in the staging waves of eight blocks.  Same (a, d), same products: against R2DM_F2_PRESPLIT_NARROW=0 bit for bit, at a batch
    where the 32-channel tiles are dispatched and at ones where they are not (then nothing changes at all)."""
    import r2dm_amd

    ck = synthetic_ckpt(resolution=res)
    x, c = rnd(9, batch, 2, *res).to(DEV), torch.linspace(-5.0, 7.0, batch).to(DEV)
    outs = {}
    saved = os.environ.get("R2DM_F2_PRESPLIT_NARROW")
    for mode in ("2", "1", "0"):  # (2: the 512 -> 512 launches of level 4 as well)
        os.environ["R2DM_F2_PRESPLIT_NARROW"] = mode
        try:
            m, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=batch)
            outs[mode] = (m.model(x, c).clone(), m.model(x, c).clone())
            sites = [n for n, _ in m.model.range_report()]
            if mode == "1" and res == (64, 1024) and batch == 8:
                assert any("operand pre-pass" in n for n in sites), sites[:5]  # (the path under test really ran)
            del m
        finally:
            os.environ.pop("R2DM_F2_PRESPLIT_NARROW", None)
            if saved is not None:
                os.environ["R2DM_F2_PRESPLIT_NARROW"] = saved
    assert torch.equal(outs["1"][0], outs["1"][1]) and torch.equal(outs["0"][0], outs["0"][1]) and torch.equal(outs["2"][0], outs["2"][1])
    assert torch.equal(outs["1"][0], outs["0"][0]) and torch.equal(outs["2"][0], outs["0"][0])


@pytest.mark.parametrize("res,batch", [((64, 1024), 8), ((32, 256), 2), ((128, 2048), 2)])
def test_fir_down_statistics_match_the_streaming_pass(res, batch):
    """Round 5: the FIR down-sampler leaves the GroupNorm statistics of its output in the convolution epilogues' slot grid (resample.hip
    fir_down2_stats_kernel; efficient_unet.py:95-97 behind :135), so the first norm of d_block2..4 needs neither the streaming pass nor -- folded
    into its consumer -- a finalize launch.  Against the same forward with the streaming pass (R2DM_FIR_STATS=0): fp64 sums in another order, so
    equal to rounding of the (a, d) pairs (not bit for bit); both repeat themselves bit for bit."""
    import r2dm_amd

    ck = synthetic_ckpt(resolution=res)
    x, c = rnd(11, batch, 2, *res).to(DEV), torch.linspace(-3.0, 5.0, batch).to(DEV)
    outs = {}
    saved = os.environ.get("R2DM_FIR_STATS")
    for mode in ("1", "0"):
        os.environ["R2DM_FIR_STATS"] = mode
        try:
            m, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=batch)
            outs[mode] = (m.model(x, c).clone(), m.model(x, c).clone())
            del m
        finally:
            os.environ.pop("R2DM_FIR_STATS", None)
            if saved is not None:
                os.environ["R2DM_FIR_STATS"] = saved
    assert torch.equal(outs["1"][0], outs["1"][1]) and torch.equal(outs["0"][0], outs["0"][1])
    assert max_abs(outs["1"][0], outs["0"][0]) < 2e-6 and rms(outs["1"][0], outs["0"][0]) < 2e-7


@pytest.mark.parametrize("res,batch", [((64, 1024), 8), ((32, 256), 3)])
def test_in_conv_channel_shares_are_bit_identical(res, batch):
    """Round 5: conv_few_in_kernel (in_conv, efficient_unet.py:262,283) splits a pixel block's output channels over 1, 2 or 4 blocks (more waves
    for the same output stream).  Every output channel is still one thread's multiply-add chain and every statistics slot one wave's sum: the
    forward is the same bit for bit whatever the split (R2DM_FEW_IN_SPLIT, read per call)."""
    import r2dm_amd

    m, _, _ = r2dm_amd.setup_model(synthetic_ckpt(resolution=res), device=DEV, show_info=False, max_batch=batch)
    x, c = rnd(13, batch, 2, *res).to(DEV), torch.linspace(-2.0, 4.0, batch).to(DEV)
    outs = []
    saved = os.environ.get("R2DM_FEW_IN_SPLIT")
    try:
        for split in ("1", "2", "4"):
            os.environ["R2DM_FEW_IN_SPLIT"] = split
            outs.append(m.model(x, c).clone())
    finally:
        os.environ.pop("R2DM_FEW_IN_SPLIT", None)
        if saved is not None:
            os.environ["R2DM_FEW_IN_SPLIT"] = saved
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.equal(outs[1], m.model(x, c))  # (the default: two shares)


def test_second_golden_resolution_32x256(golden):
    """every golden so far is 16x128 -- one 64-pixel tile column wide for most kernels.  The reference's own run at
    32x256 (tests/golden/make_golden.py res2): several tile columns and tile rows per kernel, whole denoiser at three conditions and a 4-step
    DDPM sample on the recorded noise."""
    import r2dm_amd

    g = golden("res32x256")
    res = (32, 256)
    ddpm, _, _ = r2dm_amd.setup_model(synthetic_ckpt(resolution=res), device=DEV, show_info=False)
    x = g["x"].to(DEV)
    for i, c in enumerate(g["conds"].tolist()):
        assert max_abs(ddpm.model(x, torch.full((2,), c, device=DEV)).cpu(), g["y"][i]) < 2e-5, c
    Tape(ddpm, g["sample_noise"])
    out = ddpm.sample(batch_size=2, num_steps=4, progress=False, rng=None).cpu()
    assert rms(out, g["sample_out"]) < 1e-5 and max_abs(out, g["sample_out"]) < 2e-3  # (4 coarse steps: see test_sample_golden on the tail pixels)


def test_batch8_full_size_sampler_vs_fp64_oracle_on_the_gpu():
    """the wide / tall one-accumulator tiles through the SAMPLER in the driver-run suite -- BASELINE configs[1]'s shape
    (64x1024, batch 8: the tiles conv_f16x2_pick_co_tile selects at the planned batch), 8 DDPM steps on a recorded noise tape, against the
    oracle evaluated in fp64 on this GPU (torch ops in double: no MIOpen fp32 path involved), for two samples of the batch.  Per step:
    rms(hip - fp64) <= 1e-5; the final sample within 1e-4 on all but a handful of pixels and 2e-3 at most (8 coarse steps: the tail pixels
    of test_sample_golden)."""
    import r2dm_amd
    from oracle import r2dm_oracle as O

    B, S = 8, 8
    ddpm, _, _ = r2dm_amd.setup_model(synthetic_ckpt(), device=DEV, show_info=False, max_batch=B)
    rng = r2dm_amd.setup_rng(list(range(B)), DEV)
    tape = [ddpm.randn(B, 2, 64, 1024, rng=rng, device=DEV) for _ in range(S + 1)]
    Tape(ddpm, tape)
    got = ddpm.sample(batch_size=B, num_steps=S, progress=False, rng=None, return_all=True)
    sd64 = {k: v.to(DEV).double() for k, v in O.strip_prefix(synthetic_ckpt()["ema_weights"]).items()}
    cfg = O.UNetConfig()
    worst = 0
    for i in (0, 5):  # (two samples of the batch: the fp64 oracle is slow, the samples are independent)
        want = O.sample_continuous(lambda x, c: O.unet_forward(sd64, cfg, x, c), (1, 2, 64, 1024), S, noises=[z[i:i + 1] for z in tape],
                                   return_all=True, device=DEV, dtype=torch.float64)
        rows = [(rms(got[k][i:i + 1].cpu(), want[k].cpu()), max_abs(got[k][i:i + 1].cpu(), want[k].cpu())) for k in range(S + 1)]
        print(f"batch-8 sampler, sample {i}, vs fp64 per step (rms/max): " + "  ".join(f"{a:.1e}/{b:.1e}" for a, b in rows))
        assert all(r <= 1e-5 for r, _ in rows), rows
        d = (got[-1][i:i + 1].double() - want[-1].double()).abs()
        assert d.max().item() < 2e-3 and (d > 1e-4).sum().item() <= 64, (d.max().item(), (d > 1e-4).sum().item())
        worst = max(worst, d.max().item())
    print(f"batch-8 sampler final sample: max|hip - fp64| = {worst:.2e}")


def test_event_pair_overhead_calibration(small):
    """bench.py's per-launch figures are event pairs around each convolution; r2dm_profile_event_overhead says what such a pair spans by itself
    (around nothing <= around an empty kernel, both a few microseconds), so that they can be set against rocprofv3's kernel durations."""
    a, b = small[0].model.event_pair_overhead()
    assert 0.0 < a <= b + 0.5 and b < 100.0, (a, b)
