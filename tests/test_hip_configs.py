"""-m gpu: configurations and code paths beyond the default checkpoint -- the other coordinate encodings and schedules
(SURVEY.md section 8 (f).4 / S4), layer shapes that take the engine's fallback branches, BASELINE configs[1] / configs[2]
at full size against the oracle, and the product's own multi-rank path (weight blob hand-over, two ranks on one GPU).
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import GOLDEN_RES, ROOT, max_abs, rnd, synthetic_ckpt

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(max_batch=8, **kw):
    import r2dm_amd

    ddpm, lidar, cfg = r2dm_amd.setup_model(synthetic_ckpt(**kw), device=DEV, show_info=False, max_batch=max_batch)
    return ddpm


def rms(a, b):
    return (a.double() - b.double()).pow(2).mean().sqrt().item()


def q99(a, b):
    return torch.quantile((a.double() - b.double()).abs().flatten(), 0.99).item()


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


# ---- (f).4 / S4: non-default variants against the reference's own outputs -------------------------------------------
@pytest.mark.parametrize("enc", ["spherical_harmonics", "polar_coordinates"])
def test_unet_other_coordinate_encodings(golden, enc):
    """encoding.py:92-117 / efficient_unet.py:220-226: 25 spherical-harmonics channels (in_conv 27) or the 2 raw angles
    (in_conv 4), folded into the engine's constant bias map like the Fourier features."""
    g = golden("variants")
    ddpm = build(resolution=GOLDEN_RES, coords_encoding=enc)
    y = ddpm.model(g["x"].to(DEV), g["cond"].to(DEV)).cpu()
    assert max_abs(y, g[f"unet_{enc}"]) < 2e-5


def test_linear_log_snr_schedule_p_step(golden):
    """cfg.diffusion.noise_schedule = "linear" (continuous_time.py:18-19): one teacher-forced DDPM step vs the reference."""
    g = golden("variants")
    ddpm = build(resolution=GOLDEN_RES, noise_schedule="linear")
    z = rnd(32, 2, 2, *GOLDEN_RES)
    ddpm.randn = lambda *shape, rng=None, **kw: z.to(DEV)
    y = ddpm.p_step(rnd(31, 2, 2, *GOLDEN_RES).to(DEV), torch.full((2,), 0.5), torch.full((2,), 0.375), rng=None, mode="ddpm").cpu()
    assert max_abs(y, g["linear_p_step"]) < 1e-4 and q99(y, g["linear_p_step"]) < 2e-6


# ---- engine fallback branches ------------------------------------------------------------------------
@pytest.mark.parametrize("kw", [
    dict(base_channels=32, gn_num_groups=4, attn_num_heads=4),   # in_conv Cout = 32: 32-channel fp32 tile cannot emit statistics
    dict(base_channels=96, gn_num_groups=8, attn_num_heads=12),  # 12 / 24 / 48 / 96 channels per group: no fused statistics
    dict(base_channels=64, gn_num_groups=2),                     # 32 .. 256 channels per group
], ids=["base32", "base96", "groups2"])
def test_unet_configs_off_the_fused_statistics_path(kw):
    """Shapes whose GroupNorm statistics cannot come from the producing convolution's epilogue must take the streaming
    pass (and not read an unwritten sink): U-Net output vs the fp64 oracle evaluated with torch ops on the GPU."""
    from oracle import r2dm_oracle as O

    ddpm = build(resolution=GOLDEN_RES, **kw)
    ck = synthetic_ckpt(resolution=GOLDEN_RES, **kw)
    sd64 = {k: v.double().to(DEV) for k, v in O.strip_prefix(ck["ema_weights"]).items()}
    cfg = O.UNetConfig(resolution=GOLDEN_RES, base_channels=kw.get("base_channels", 64), gn_num_groups=kw.get("gn_num_groups", 8),
                       attn_num_heads=kw.get("attn_num_heads", 8))
    x, cond = rnd(90, 2, 2, *GOLDEN_RES), torch.tensor([-3.0, 5.0])
    y = ddpm.model(x.to(DEV), cond.to(DEV))
    ref = O.unet_forward(sd64, cfg, x.double().to(DEV), cond.double().to(DEV))
    assert max_abs(y.cpu(), ref.cpu()) < 2e-5
    assert torch.equal(y, ddpm.model(x.to(DEV), cond.to(DEV)))


def test_random_geometries_against_the_fp64_oracle():
    """Round 3 fuzz (scripts/fuzz_configs.py; 200 random cases in profiles/r03_fuzz.txt): random resolutions (down to 8x64: level 4 is
    1x8), base widths 16 ... 128, channel multipliers, block counts, group counts, head counts, batch sizes and precision modes; every
    case is compared with the float64 oracle on the device, must repeat bit-identically and sample finitely.  The fuzz found the one
    out-of-bounds read of this library (a residual plane beyond Cout when a 32-channel tile covers a 16-channel layer: a memory fault
    only when the tensor ends a mapped segment) and made the generic attention kernel necessary (head sizes other than 32 / 64, token
    counts that are not a multiple of 32).  A dozen cases ride in the suite."""
    env = dict(os.environ, SEED="11", CASES="12")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_configs.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "12 cases, 0 failures" in r.stdout and " rejected" not in r.stdout, r.stdout[-3000:]


def test_random_sampler_arguments_against_the_oracle():
    """scripts/fuzz_sampler.py: random objective / mode / eta / schedule / step count / batch (also beyond max_batch) / RNG form / return_all
    against the oracle's sampler on the same generators (profiles/r03_fuzz.txt); the reference's exceptions for a bad mode and a generator
    list of the wrong length."""
    env = dict(os.environ, SEED="5", CASES="10")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_sampler.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "10 cases, 0 failures" in r.stdout and r.stdout.count("Error OK") == 2, r.stdout[-3000:]


def test_ddim_eta1_with_the_schedule_scalars_on_the_device():
    """VERDICT round 5, missing #3 / item 6: the reference evaluates linspace, log-SNR, alpha, sigma on `self.device`
    (/root/reference/models/diffusion/continuous_time.py:203-206,248-249); the product's default is the host (pinned to the reference's CPU
    run).  For DDIM with eta = 1, c_2 = sqrt(1 - alpha_s^2 - c_1^2) is a rounding residue on the first and last step, and the two libms put
    the sample 5e-4 apart (profiles/r05_fuzz.txt, case 15).  `schedule_on="device"`: (1) the table's rows ARE the oracle's device-evaluated
    scalars, bit for bit; (2) the eta = 1 sample then meets the tight bar against the oracle that evaluates its scalars on the device."""
    import r2dm_amd
    from oracle import r2dm_oracle as O
    from r2dm_amd import synthetic

    dev = torch.device("cuda", 0)
    RES, S, B, ETA = (16, 128), 4, 4, 1.0
    ck = synthetic.synthetic_checkpoint(seed=0, resolution=RES, prediction_type="eps", noise_schedule="cosine")
    ddpm, _, _ = r2dm_amd.setup_model(ck, device=dev, show_info=False, max_batch=8, schedule_on="device")
    assert ddpm.schedule_on == "device"
    # (1) the scalars: the table against the reference's expressions on (B,)-shaped device tensors, step by step
    cond, coef, _ = ddpm._sample_tables(S, B, "ddim", ETA, dev)
    steps = torch.linspace(1.0, 0.0, S + 1, device=dev)[None].repeat_interleave(B, dim=0)
    for i in range(S):
        lt, ls = O.log_snr_cosine(steps[:, i]), O.log_snr_cosine(steps[:, i + 1])
        a_t, s_t = O.alpha_sigma(lt)
        a_s, s_s = O.alpha_sigma(ls)
        c1 = ETA * s_s / s_t * (1 - a_t**2 / a_s**2).sqrt()
        c2 = (1 - a_s**2 - c1**2).sqrt()
        assert torch.equal(cond[i], lt)
        assert torch.equal(coef[i], torch.stack([a_t, s_t, a_s, s_s, torch.zeros_like(lt), torch.zeros_like(lt), c1, c2], dim=-1))
    # ... and p_step's (B,)-shaped evaluation agrees with the table
    _, krow, _ = ddpm._coefficients(steps[:, 0], steps[:, 1], "ddim", ETA)
    assert torch.equal(krow, coef[0])
    # (2) the sample
    mk = lambda: r2dm_amd.setup_rng(list(range(100, 100 + B)), dev)
    got = ddpm.sample(batch_size=B, num_steps=S, progress=False, rng=mk(), mode="ddim", ddim_eta=ETA)
    sd = {k: v.double().to(dev) for k, v in O.strip_prefix(ck["ema_weights"]).items()}
    cfg = O.UNetConfig(resolution=RES)
    net = lambda x, c: O.unet_forward(sd, cfg, x.double(), c.double()).float()
    g = mk()
    tape = [O.draw_noise((B, 2, *RES), g, dev, torch.float32) for _ in range(S + 1)]
    want = O.sample_continuous(net, (B, 2, *RES), S, noises=tape, mode="ddim", ddim_eta=ETA, objective="eps", device=dev)
    e = (got - want).abs().flatten().double()
    eq99, erms, emax = torch.quantile(e, 0.99).item(), e.pow(2).mean().sqrt().item(), e.max().item()
    # the host-evaluated table on the same tape, for the record (the 5e-4 of the fuzz's case 15)
    ddpm.set_schedule_on("host")
    host = ddpm.sample(batch_size=B, num_steps=S, progress=False, rng=mk(), mode="ddim", ddim_eta=ETA)
    print(f"DDIM eta=1: |hip(schedule on device) - oracle(device scalars)| q99 {eq99:.2e} rms {erms:.2e} max {emax:.2e}; "
          f"hip(schedule on host) vs the same oracle: q99 {q99(host, want):.2e} rms {rms(host, want):.2e}")
    # tight bar: what DDPM / DDIM eta < 1 meet (profiles/r05_fuzz.txt: q99 <= 5e-6, rms <= 2e-5; the max is the ill-conditioned clamp tail)
    assert eq99 < 5e-6 and erms < 2e-5 and emax < 3e-3, (eq99, erms, emax)


def test_unsupported_channel_multiplier_is_rejected():
    """An up stage whose concatenated input equals its output width would take an identity skip over a concatenation."""
    import r2dm_amd
    from r2dm_amd import _lib

    ck = synthetic_ckpt(resolution=GOLDEN_RES, channel_multiplier=(2, 1, 2, 4))
    ddpm, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False)
    with pytest.raises(_lib.R2DMError, match="identity"):
        ddpm.model(torch.zeros(1, 2, *GOLDEN_RES, device=DEV), torch.zeros(1, device=DEV))


# ---- BASELINE configs[1] / configs[2] at full size ------------------------------------------------------------------------
def test_batch8_full_size_forward_vs_fp64_oracle():
    """configs[1] geometry AND batch: the layer tilings are chosen for max_batch (batch 8 takes other variants than batch
    1 for several layers).  One sample of the batch against the fp64 oracle (torch ops on the GPU); the batch row equals
    the same sample run alone through the same engine."""
    from oracle import r2dm_oracle as O

    ddpm = build(max_batch=8)
    sd64 = {k: v.double().to(DEV) for k, v in O.strip_prefix(synthetic_ckpt()["ema_weights"]).items()}
    cfg = O.UNetConfig()
    x = rnd(91, 8, 2, 64, 1024)
    cond = torch.linspace(-12.0, 12.0, 8)
    y = ddpm.model(x.to(DEV), cond.to(DEV))
    for i in (0, 5):
        ref = O.unet_forward(sd64, cfg, x[i:i + 1].double().to(DEV), cond[i:i + 1].double().to(DEV))
        e = max_abs(y[i:i + 1].cpu(), ref.cpu())
        print(f"batch-8 forward, sample {i}: max|hip - fp64| = {e:.2e}")
        assert e < 8e-6  # measured 2.1e-6
    assert torch.equal(y[5:6], ddpm.model(x[5:6].to(DEV), cond[5:6].to(DEV)))


def test_ddim_32_step_final_sample_parity_full_size():
    """BASELINE configs[2]: 64x1024, 32-step DDIM (eta 0), the FINAL sample on one noise tape against the oracle -- the
    reference's arithmetic (torch fp32 on the CPU) -- and both against fp64 'exact arithmetic' of the same algorithm.

    DDIM draws no fresh noise, so the amplified roundoff of the first steps (x_0 = (x_t - sigma eps)/alpha_t with
    1/alpha_t ~ 1800 at t = 1, see test_hip_unet.test_sample_golden) is carried to the end instead of being contracted:
    the reference itself ends 5.5e-4 from exact arithmetic at its worst pixel (rms 4.2e-6, q99 1.1e-6), i.e. BASELINE's
    "per-pixel delta < 1e-4" is below the reference's own fp32 floor for this sampler.  Measured here: HIP vs exact max
    1.4e-3, rms 7.4e-6, q99 1.8e-6; HIP vs reference max 1.5e-3, rms 7.6e-6.  The statement that holds, and is asserted:
    the HIP sampler's error against exact arithmetic is within 2x of the reference's in RMS and in the 99th percentile
    (the U-Net's convolutions accumulate with ~1.7x the roundoff of oneDNN's blocked fp32 sums, profiles/
    r02a_error_budget_default.txt), within 4x at the worst pixel (a lottery over a handful of unclamped pixels at t ~ 1)."""
    import r2dm_amd
    from oracle import r2dm_oracle as O

    S = 32
    ddpm = build(max_batch=1)
    rng = r2dm_amd.setup_rng([21], DEV)
    tape = [ddpm.randn(1, 2, 64, 1024, rng=rng, device=DEV) for _ in range(S + 1)]
    Tape(ddpm, tape)
    got = ddpm.sample(batch_size=1, num_steps=S, progress=False, rng=None, mode="ddim").cpu()
    sd = O.strip_prefix(synthetic_ckpt()["ema_weights"])
    cfg = O.UNetConfig()
    cpu_tape = [z.cpu() for z in tape]
    want = O.sample_continuous(lambda x, c: O.unet_forward(sd, cfg, x, c), (1, 2, 64, 1024), S, noises=cpu_tape, mode="ddim")
    sd64 = {k: v.double().to(DEV) for k, v in sd.items()}
    truth = O.sample_continuous(lambda x, c: O.unet_forward(sd64, cfg, x, c), (1, 2, 64, 1024), S, noises=tape, mode="ddim",
                                device=DEV, dtype=torch.float64).cpu()
    row = dict(max_hip_ref=max_abs(got, want), rms_hip_ref=rms(got, want), max_hip_truth=max_abs(got, truth), rms_hip_truth=rms(got, truth),
               q99_hip_truth=q99(got, truth), max_ref_truth=max_abs(want, truth), rms_ref_truth=rms(want, truth), q99_ref_truth=q99(want, truth))
    print("DDIM 32-step final sample 64x1024: " + "  ".join(f"{k} {v:.2e}" for k, v in row.items()))
    assert row["rms_hip_truth"] <= max(2 * row["rms_ref_truth"], 2e-6), row
    assert row["q99_hip_truth"] <= max(2 * row["q99_ref_truth"], 2e-6), row
    assert row["max_hip_truth"] <= max(4 * row["max_ref_truth"], 1e-4), row
    assert row["rms_hip_ref"] < 2e-5 and row["max_hip_ref"] < 5e-3, row


# ---- the product's multi-rank path ----------------------------------------------------------------------------------------
def test_adopted_weight_blob_reproduces_the_packing_model():
    """What every rank but 0 does in a multi-GPU run (distributed.py:38-50, unet.py adopt_packed_weights): bind the
    broadcast blob instead of packing its own weights.  Model B holds DIFFERENT parameters; after adopting A's blob its
    forward is bit-identical to A's."""
    import r2dm_amd
    from r2dm_amd import synthetic

    a = build(resolution=GOLDEN_RES)
    ck_b = synthetic.synthetic_checkpoint(seed=1, resolution=GOLDEN_RES)
    b, _, _ = r2dm_amd.setup_model(ck_b, device=DEV, show_info=False)
    x, cond = rnd(92, 3, 2, *GOLDEN_RES).to(DEV), torch.tensor([-2.0, 0.5, 9.0], device=DEV)
    ya, yb = a.model(x, cond), b.model(x, cond)
    assert not torch.equal(ya, yb)
    blob = a.model.packed_weights(DEV).clone()  # (a broadcast delivers a copy)
    assert blob.numel() == b.model.packed_weight_bytes()
    b.model.adopt_packed_weights(blob)
    assert torch.equal(b.model(x, cond), ya)
    fresh, _, _ = r2dm_amd.setup_model(ck_b, device="cpu", show_info=False)  # a rank that never packed anything
    fresh.to(DEV)
    fresh.model.adopt_packed_weights(blob, layout_hash=a.model.packed_layout_hash())
    assert torch.equal(fresh.model(x, cond), ya)
    # (a blob planned for another batch size is refused: tests/test_host.py::test_blob_layout_fingerprint)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("cfg,batch", [(1, 2), (4, 1)])  # BASELINE configs[1] / configs[3] geometry, and configs[4]'s 128x2048
def test_two_rank_bench_on_one_gpu_matches_single_process(tmp_path, cfg, batch):
    """bench.py under torch.distributed.run with two ranks sharing this GPU (gloo instead of RCCL): rank 0 packs, the blob
    is broadcast, rank 1 adopts it; every rank samples its own seed shard.  Each rank's samples equal those of a
    single-process run of the same seeds (sample_and_save.py:37-46,75: partition invariance).

    All runs use the DEFAULT kernels.  Round 2 ran this test with R2DM_CONV_ALGO=f32 because forwards next to a second process came
    out wrong (96 of 150) and blamed the LDS-DMA kernels; round 3 bisected it (scripts/jobs/j74-j77.sh, profiles/r03_shared_gpu.txt,
    LABNOTES.md section 6): the failure belonged to the old LDS-tiled out_conv kernel, disappeared with the commit that replaced it
    (1c7a0dc) and does not occur with any kernel of the current library -- 0 of 700 forwards at batch 2 / 8, both operand splits."""
    env = dict(os.environ, R2DM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--config", str(cfg), "--steps", "2", "--warmup", "1", "--batch", str(batch), "--no-cpu-baseline", "--no-torch-baseline", "--no-exact-baseline", "--no-other-configs",
              "--prewarm-s", "0.5"]
    d2 = tmp_path / "two"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dump-samples", str(d2)] + common,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 2 * batch and line["scaling"] == "weak"
    # the line proves the collective ran: backend, world size, and one all-reduce of the adopted blob's checksum (equal on all ranks)
    assert line["rccl"]["world_size"] == 2 and line["rccl"]["backend"].startswith("gloo") and line["rccl"]["blob_crc_equal_on_all_ranks"] is True
    for rank in (0, 1):
        d1 = tmp_path / f"one{rank}"
        r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--seed-base", str(batch * rank), "--dump-samples", str(d1)] + common,
                            env=env, capture_output=True, text=True, timeout=900)
        assert r1.returncode == 0, r1.stderr[-2000:]
        two, one = torch.load(d2 / f"rank{rank}.pt"), torch.load(d1 / "rank0.pt")
        assert two["seeds"] == one["seeds"] == list(range(batch * rank, batch * rank + batch))
        assert torch.equal(two["samples"], one["samples"])


def test_bench_gpus_2_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (re-exec under torch.distributed.run);
    under gloo they share this GPU and the line reports n_gpus 2 and a world of 2.  With the RCCL backend the same command on a one-GPU
    box must stop in seconds with a message naming the device count (RCCL wants one device per rank)."""
    common = ["--steps", "2", "--warmup", "1", "--batch", "1", "--no-cpu-baseline", "--no-torch-baseline", "--no-exact-baseline", "--no-other-configs", "--prewarm-s", "0.2"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, env=dict(env, R2DM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl"]["world_size"] == 2 and line["config"]["global_batch"] == 2
    assert line["dtype"] == "f32" and line["dtype_note"].startswith("22-bit split fp16 operands") and line["roofline"]["traffic_measured"] is False
    if torch.cuda.device_count() == 1:
        import time
        t0 = time.time()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, env=dict(env, R2DM_DIST_BACKEND="nccl"), capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "device_count() = 1" in r.stderr and time.time() - t0 < 120, r.stderr[-500:]


_RCCL_SELFTEST = r"""
import os, sys, torch
import torch.distributed as td
sys.path.insert(0, sys.argv[1])
import r2dm_amd
from r2dm_amd import synthetic
from r2dm_amd.distributed import sample_sharded
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
td.init_process_group("nccl", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=0, world_size=1, device_id=dev)  # "nccl" IS RCCL on ROCm
res = (16, 128)
a, _, _ = r2dm_amd.setup_model(synthetic.synthetic_checkpoint(seed=0, resolution=res), device=dev, show_info=False, max_batch=2)
b, _, _ = r2dm_amd.setup_model(synthetic.synthetic_checkpoint(seed=1, resolution=res), device=dev, show_info=False, max_batch=2)
wire = torch.empty_like(a.model.packed_weights(dev))
wire.copy_(a.model.packed_weights(dev))
td.broadcast(wire, src=0)                      # the blob through an RCCL broadcast kernel
b.model.adopt_packed_weights(wire)
x = torch.randn(2, 2, *res, device=dev); c = torch.tensor([0.3, -2.0], device=dev)
assert torch.equal(a.model(x, c), b.model(x, c))
t = torch.ones(4, device=dev); td.all_reduce(t); torch.cuda.synchronize(); assert t.sum().item() == 4.0
out, mine = sample_sharded(lambda seeds: a.sample(batch_size=len(seeds), num_steps=2, progress=False, rng=r2dm_amd.setup_rng(seeds, dev)), [5, 6])
assert mine == [5, 6] and out.shape[0] == 2
td.barrier(); td.destroy_process_group()
print("RCCL_SELFTEST_OK", td.is_nccl_available())
"""


def test_rccl_backend_executes_on_this_gpu():
    """The multi-GPU path uses torch.distributed's "nccl" backend (= RCCL on ROCm) for one broadcast of the packed weight blob
    (r2dm_amd/distributed.py; /root/reference/sample_and_save.py:25-46 re-reads the checkpoint on every rank instead).  The GPU
    boxes this suite runs on have ONE device, so the N-rank form cannot run here (gloo world-size-2: tests/test_distributed_cpu.py,
    test_two_rank_bench_on_one_gpu...); this test at least executes the RCCL communicator set-up, a broadcast of the real blob, an
    all-reduce and a barrier on the hardware with a one-rank group, and checks that a model adopting the broadcast blob
    reproduces the packing model bit for bit."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_SELFTEST, ROOT, str(_free_port())], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_SELFTEST_OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.skipif(os.environ.get("R2DM_CONV_ALGO", "").startswith("f"), reason="fp32-MFMA algorithm forced")
@pytest.mark.parametrize("batch,res,reps,neighbour,modes", [
    (2, (64, 1024), 80, "512,512,8,128,1,8", ("fp32", "fp32-bf16x3", "fp16")),
    (8, (64, 1024), 40, "512,512,8,128,1,8", ("fp32", "fp32-bf16x3", "fp16")),
    (2, (128, 2048), 40, "512,512,8,128,1,8", ("fp32", "fp32-bf16x3", "fp16")),
    (2, (64, 1024), 600, "64,2,64,1024,3,8", ("fp16",)),  # round 5: the trigger of the one-plane 32-channel tile's fault (17 of 20 such runs)
])
def test_forwards_next_to_a_second_process(tmp_path, batch, res, reps, neighbour, modes):
    """Round 2 saw wrong forwards whenever a second process computed on the same GPU; round 3 traced it to ONE kernel (the old
    LDS-tiled out_conv, now removed: profiles/r03_shared_gpu.txt) and found the most effective trigger: a second process
    looping a 55 KB-LDS GEMM-like kernel (147 of 150 forwards wrong with the old kernel).  Every precision mode of the
    current library next to that neighbour: 0 forwards may differ bitwise from the one computed alone
    (/root/reference/sample_and_save.py:37-46,75: results must not depend on the process layout).  Round 4: also at
    batch 8 (BASELINE configs[1]: the 128-channel-tile kernels run there) and on the per-GPU shard of configs[4] (128x2048, batch 2).
    Round 5: next to a looping out_conv (the direct kernel: 133 registers, no LDS -- it fits beside any block that leaves registers free)
    the one-plane 32-channel tile of conv_f16x2.hip died of a GPU memory fault in 17 of 20 runs of 600 fp16-mode forwards; since a
    conv_f16x2 block owns its CU, 0 of 70 (profiles/r05_coresidency.txt) -- the last case repeats that run."""
    import time

    ready = tmp_path / "ready"
    env = dict(os.environ, SHAPE=neighbour, SECS="150", READY_FILE=str(ready))
    ddpm = build(max_batch=batch, resolution=res)
    x, c = rnd(91, batch, 2, *res).to(DEV), torch.linspace(-3.0, 1.0, batch, device=DEV)
    alone = {}
    for m in modes:
        ddpm.model.set_precision(m)
        alone[m] = ddpm.model(x, c).clone()
    hog = subprocess.Popen([sys.executable, os.path.join(ROOT, "scripts", "hog_conv_loop.py")], env=env, cwd=ROOT,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        t0 = time.time()
        while not ready.exists() and time.time() - t0 < 120 and hog.poll() is None:
            time.sleep(0.5)
        assert ready.exists(), "the neighbour process did not come up"
        bad = {}
        for m in modes:
            ddpm.model.set_precision(m)
            with ddpm.model.deferred_range_check():
                bad[m] = sum(int(not torch.equal(ddpm.model(x, c), alone[m])) for _ in range(reps))
        assert hog.poll() is None, "the neighbour exited before the comparison ended"
    finally:
        hog.terminate()
        hog.wait()
    print("forwards differing next to a second process:", bad)
    assert bad == {m: 0 for m in modes}


def test_compile_and_autocast_wrapping_degrades_to_the_same_eager_call():
    """sample_and_save.py:14,45,70 upstream: ddpm wrapped by torch.compile (dynamo errors suppressed) and sampled under
    fp16 autocast.  The denoiser's forward is one ctypes call into libr2dm_hip.so -- a graph break that runs eagerly --
    and the engine computes in fp32 whatever the autocast state: same bits as the plain model."""
    import torch._dynamo

    import r2dm_amd

    torch._dynamo.config.suppress_errors = True
    ck = synthetic_ckpt(resolution=GOLDEN_RES)
    plain, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False)
    wrapped, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, compile=True)
    a = plain.sample(batch_size=2, num_steps=3, progress=False, rng=r2dm_amd.setup_rng([0, 1], DEV))
    with torch.autocast("cuda", dtype=torch.float16):
        b = wrapped.sample(batch_size=2, num_steps=3, progress=False, rng=r2dm_amd.setup_rng([0, 1], DEV))
    assert b.dtype == torch.float32 and torch.equal(a, b)


# (the GroupNorm-gain range tests live in tests/test_hip_range.py: the guard follows the observed maxima since round 3)


@pytest.mark.skipif(os.environ.get("R2DM_CONV_ALGO", "").startswith("f"), reason="fp32-MFMA algorithm forced")
def test_f16x2_skip_convolution_input_range_fails_loudly():
    """The 1x1 skip convolution of an up block reads the raw concatenation [previous up stage | skip tensor] on the fp16
    matrix pipe (proj_f16x2.hip): both producers record max|output|.  An up-sampling convolution whose bias pushes its output
    out of the fp16 range (the GroupNorm behind it would normalise that away) must make the forward fail; the bf16x3 mode
    runs the same weights."""
    import r2dm_amd
    from r2dm_amd._lib import R2DMError

    ck = dict(synthetic_ckpt())
    ck["ema_weights"] = dict(ck["ema_weights"])
    key = next(k for k in ck["ema_weights"] if k.endswith("u_block4.upsample.1.bias"))
    ck["ema_weights"][key] = torch.full_like(ck["ema_weights"][key], 1.0e5)
    ddpm, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=2, strict_range=True)
    x, c = rnd(97, 2, 2, 64, 1024).to(DEV), torch.zeros(2, device=DEV)
    with pytest.raises(R2DMError, match="fp16 range"):
        ddpm.model(x, c)
    ddpm.model.set_precision("fp32-bf16x3")
    assert torch.isfinite(ddpm.model(x, c)).all()


@pytest.mark.skipif(os.environ.get("R2DM_CONV_ALGO", "").startswith("f"), reason="fp32-MFMA algorithm forced")
def test_f16x2_tracked_activation_range_fails_loudly():
    """The down- and up-sampling convolutions have no GroupNorm in front: on the f16x2 path their inputs' running maximum is
    recorded by the producing kernel (conv epilogue / fir_up2).  An input scaled so that the residual stream leaves the
    fp16 range must make the forward fail; the bf16x3 mode computes it (finite, and equal to the reference)."""
    from r2dm_amd._lib import R2DMError

    ddpm = build(max_batch=2)
    ddpm.model.strict_range = True  # (the default falls back to the wide-range split: tests/test_hip_range.py)
    x, c = rnd(96, 2, 2, 64, 1024).to(DEV), torch.zeros(2, device=DEV)
    assert torch.isfinite(ddpm.model(x, c)).all()  # ordinary inputs: no complaint
    with pytest.raises(R2DMError, match="fp16 range"):
        ddpm.model(x * 3e6, c)
    assert torch.isfinite(ddpm.model(x, c)).all()  # the flag is per forward
    ddpm.model.set_precision("fp32-bf16x3")
    assert torch.isfinite(ddpm.model(x * 3e6, c)).all()
