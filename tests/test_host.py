"""CPU tests: host logic, state-dict compatibility, the C-ABI surface, loud failure without a GPU."""
import ctypes
import os
import re

import pytest
import torch

from conftest import GOLDEN_RES, ROOT, synthetic_ckpt


def test_library_exports_every_declared_symbol():
    from r2dm_amd import _lib

    header = open(os.path.join(ROOT, "include", "r2dm_hip.h")).read()
    declared = set(re.findall(r"\b(r2dm_[a-z0-9_]+)\s*\(", header))
    declared -= {"r2dm_handle"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.r2dm_version()


def test_fir_down_statistics_geometry_host_logic():
    """r2dm_fir_down2_stat_slots (resample.hip, host side): where the down-sampler leaves the GroupNorm statistics of its output -- the three
    down-samplers of the 64x1024 and 128x2048 networks (efficient_unet.py:135 in front of :95-97) and of the 32x256 golden geometry -- and where
    it must refuse, so that the engine falls back to the streaming pass: too few patches per slot, channels that do not split into 8 groups of
    8 / 16 / 32 / 64, widths / heights the wide kernel does not take.  The slot count is the convolution epilogues' (conv_stat_slots of the
    OUTPUT geometry: 4-row x 64-pixel tiles x 4 rows x 2 halves)."""
    from r2dm_amd import _lib

    f = _lib.lib().r2dm_fir_down2_stat_slots

    def epi_slots(h, w):
        return ((h + 3) // 4) * ((w + 63) // 64) * 4 * 2

    for res in ((64, 1024), (128, 2048), (32, 256)):
        h, w = res
        for level, c in enumerate((128, 256, 512)):
            hh, ww = h >> level, w >> level  # the down-sampler's INPUT geometry at this level
            assert f(c, 8, hh, ww) == epi_slots(hh // 2, ww // 2), (res, level)
    os.environ["R2DM_FIR_STATS"] = "0"
    try:
        assert f(128, 8, 64, 1024) == 0  # (the A/B switch)
    finally:
        del os.environ["R2DM_FIR_STATS"]
    assert f(512, 8, 4, 32) == 0      # 16x128 golden network, level 3: 32 patches per slot (< one wave's 64)
    assert f(64, 8, 8, 64) == 0       # fewer than 64 patches per slot
    assert f(128, 8, 64, 1020) == 0   # width not a multiple of 8
    assert f(128, 8, 62, 1024) == 0   # height not a multiple of 4
    assert f(96, 8, 64, 1024) == 0    # 12 channels per group
    assert f(2048, 8, 64, 1024) == 0  # 256 channels per group
    assert f(128, 7, 64, 1024) == 0   # channels not divisible by the groups


def test_engine_plan_and_blob_layout_without_gpu():
    """Handle creation, tensor table and workspace sizing are host-only."""
    from r2dm_amd import _lib, synthetic
    from r2dm_amd.unet import _Engine

    g = synthetic.geometry_from_cfg(synthetic.default_cfg_dict())
    eng = _Engine(g, 8)
    slots = list(eng.slots())
    keys = [k for _, k, _ in slots]
    sd = synthetic.synthetic_state_dict(g, prefix="")
    fixed = {k for k in sd if k.endswith(".kernel") or k in ("coords", "coords_encoding.freqs", "coords_encoding.phase")}
    assert set(keys) == (set(sd) - fixed) | {"__cenc", "__sin_freqs"}
    for _, k, n in slots:
        if not k.startswith("__"):
            assert sd[k].numel() == n, k
    assert sum(n for _, k, n in slots if not k.startswith("__")) >= 31_099_650 - 2 * 64 * 1024 - 64
    # ~31 M floats + padding; split-bf16 conv weights take 6 bytes per value instead of 4, the residual blocks' 3x3
    # weights are also kept in the f16x2 packing (4 bytes per value; the mode is switchable per handle); + the constant
    # (64, H, W) in_conv map of the Fourier channels (16.8 MB)
    assert 250e6 < eng.blob_bytes() < 330e6
    L = _lib.lib()
    w1, w8 = L.r2dm_workspace_bytes(eng.h, 1), L.r2dm_workspace_bytes(eng.h, 8)
    assert 0 < w1 < w8 < 4e9 and abs(w8 / w1 - 8) < 1.0


def test_bad_geometry_is_refused():
    from r2dm_amd import _lib
    from r2dm_amd.spec import UNetGeometry
    from r2dm_amd.unet import _Engine

    with pytest.raises(_lib.R2DMError, match="attention"):
        _Engine(UNetGeometry.make(2, (16, 128), base_channels=64, attn_num_heads=3), 1)  # 512 / 256 channels do not split into 3 heads
    _Engine(UNetGeometry.make(2, (16, 128), base_channels=8), 1)  # head size 8 / 4: the generic attention kernel (round 3; was refused)
    with pytest.raises(_lib.R2DMError):
        _Engine(UNetGeometry.make(2, (20, 100), base_channels=64), 1)


def test_state_dict_layout_matches_reference_listing():
    """268 keys, the reference's names; strict load both ways."""
    import r2dm_amd

    ck = synthetic_ckpt(resolution=GOLDEN_RES)
    ddpm, lidar, cfg = r2dm_amd.setup_model(ck, device="cpu", show_info=False)
    sd = ddpm.state_dict()
    assert len(sd) == 268 and list(sd)[0] == "_dummy"
    assert set(sd) == set(ck["ema_weights"])
    for k, v in sd.items():
        assert torch.equal(v, ck["ema_weights"][k]), k
    assert sum(p.numel() for p in r2dm_amd.setup_model(synthetic_ckpt(), show_info=False)[0].parameters()) == 31_099_650
    assert ddpm.sampling_shape == (2, *GOLDEN_RES) and ddpm.device.type == "cpu"
    assert ddpm.model.coords.shape == (1, 2, *GOLDEN_RES)
    bad = dict(ck["ema_weights"])
    bad.pop("model.out_conv.bias")
    with pytest.raises(RuntimeError):
        ddpm.load_state_dict(bad)
    assert lidar.ray_angles.shape == (1, 2, *GOLDEN_RES) and cfg.data.max_depth == 80.0


def test_no_cpu_fallback():
    import r2dm_amd
    from r2dm_amd._lib import R2DMError

    ddpm, _, _ = r2dm_amd.setup_model(synthetic_ckpt(resolution=GOLDEN_RES), device="cpu", show_info=False)
    with pytest.raises(R2DMError, match="no CPU fallback"):
        ddpm.sample(batch_size=1, num_steps=1, progress=False)
    with pytest.raises(R2DMError, match="no CPU fallback"):
        ddpm.model(torch.zeros(1, 2, *GOLDEN_RES), torch.zeros(1))
    with pytest.raises(NotImplementedError):
        ddpm(torch.zeros(1, 2, *GOLDEN_RES))  # training loss is out of scope


def test_refinenet_config_fails_like_the_reference():
    """SURVEY.md section 8(f).4, last part: the reference's own setup_model cannot build architecture='refinenet'
    (utils/inference.py:36 rebinds `in_channels` to an int, :54 calls sum() on it -> TypeError; verified by running the
    reference on such a config in the dev container).  The drop-in raises the same error type with the same message head;
    an unknown architecture raises the reference's ValueError("Unknown: ...") (utils/inference.py:60)."""
    import copy

    import r2dm_amd

    ck = copy.deepcopy(synthetic_ckpt(resolution=GOLDEN_RES))
    ck["cfg"]["model"]["architecture"] = "refinenet"
    with pytest.raises(TypeError, match="'int' object is not iterable"):
        r2dm_amd.setup_model(ck, device="cpu", show_info=False)
    with pytest.raises(NotImplementedError, match="refinenet"):  # (callers that caught round 2's error type keep working)
        r2dm_amd.setup_model(ck, device="cpu", show_info=False)
    ck["cfg"]["model"]["architecture"] = "unet3"
    with pytest.raises(ValueError, match="Unknown: unet3"):
        r2dm_amd.setup_model(ck, device="cpu", show_info=False)


def test_precision_names():
    """`precision=` of setup_model / EfficientUNet.set_precision: two parity modes, the reduced fp16 bulk mode, round 1's
    name as a deprecated alias; nothing else."""
    import r2dm_amd

    ddpm, _, _ = r2dm_amd.setup_model(synthetic_ckpt(resolution=GOLDEN_RES), device="cpu", show_info=False, precision="fp16")
    net = ddpm.model
    assert net.precision == "fp16" and sorted(net.PRECISIONS) == ["fp16", "fp32", "fp32-bf16x3"]
    with pytest.warns(DeprecationWarning):
        net.set_precision("bf16x2")
    assert net.precision == "fp32"
    with pytest.raises(ValueError):
        net.set_precision("int8")


def test_coefficient_tables_match_reference_schedule(golden):
    """Host-side schedule scalars are bit-identical to what the reference computes for batch 1 (golden 'schedule')."""
    from r2dm_amd import diffusion as D

    class Stub(torch.nn.Module):
        resolution, in_channels = GOLDEN_RES, 2

    d = D.ContinuousTimeGaussianDiffusion(Stub())
    g = golden("schedule")
    # the product evaluates every row on 1-element tensors (scalar libm path), like the reference's sample()
    # does for batch 1; the golden schedule was captured the same way -> bit-identical
    for S in (8, 32, 256):
        t = torch.linspace(1.0, 0.0, S + 1)
        cond, coef, mode = d._coefficients(t[:-1], t[1:], "ddpm", 0.0)
        assert torch.equal(cond, g[f"lam{S}"][:-1])
        assert torch.equal(coef[:, 0], g[f"alpha{S}"][:-1]) and torch.equal(coef[:, 1], g[f"sigma{S}"][:-1])
        assert torch.equal(coef[:, 2], g[f"alpha{S}"][1:]) and torch.equal(coef[:, 3], g[f"sigma{S}"][1:])
        assert torch.equal(coef[:, 4], g[f"c{S}"]) and mode == 0
        assert torch.equal(coef[:, 5], torch.cat([g[f"sigma{S}"][i + 1:i + 2] * g[f"c{S}"][i:i + 1].sqrt() for i in range(S)]))
        # and it does not depend on how many rows are evaluated together
        c1, k1, _ = d._coefficients(t[3:4], t[4:5], "ddpm", 0.0)
        assert torch.equal(c1, cond[3:4]) and torch.equal(k1, coef[3:4])
    # sample() keeps its last tables (same schedule + step count + batch + mode -> the same device tensors, nothing rebuilt);
    # anything that enters the table is part of the key
    c8, k8, m8 = d._sample_tables(8, 2, "ddpm", 0.0, "cpu")
    t8 = torch.linspace(1.0, 0.0, 9)
    r8 = d._coefficients(t8[:-1], t8[1:], "ddpm", 0.0)
    assert torch.equal(c8, r8[0][:, None].expand(8, 2)) and torch.equal(k8, r8[1][:, None, :].expand(8, 2, 8)) and m8 == r8[2]
    assert d._sample_tables(8, 2, "ddpm", 0.0, "cpu")[1] is k8
    assert d._sample_tables(8, 3, "ddpm", 0.0, "cpu")[1].shape == (8, 3, 8)
    assert not torch.equal(d._sample_tables(8, 2, "ddim", 0.0, "cpu")[1], k8)
    assert d._sample_tables(32, 2, "ddpm", 0.0, "cpu")[1].shape == (32, 2, 8)
    for S in range(40, 52):  # (bounded: old entries leave)
        d._sample_tables(S, 1, "ddpm", 0.0, "cpu")
    assert len(d.__dict__["_tables"]) <= 8
    with pytest.raises(ValueError, match="invalid mode"):
        d._coefficients(t[:-1], t[1:], "euler", 0.0)
    with pytest.raises(ValueError):
        D.ContinuousTimeGaussianDiffusion(Stub(), noise_schedule="nope")
    bad = D.ContinuousTimeGaussianDiffusion(Stub(), prediction_type="score")
    with pytest.raises(ValueError, match="invalid objective"):
        bad.sample(1, 1, progress=False)


def test_rng_draw_shapes_and_order():
    """randn semantics of base.py:71-94: None / Generator / list of Generators (len == B)."""
    from r2dm_amd import diffusion as D, setup_rng

    class Stub(torch.nn.Module):
        resolution, in_channels = (4, 8), 2

    d = D.ContinuousTimeGaussianDiffusion(Stub())
    gens = setup_rng([5, 6], "cpu")
    a = d.randn(2, 2, 4, 8, rng=gens)
    assert torch.equal(a[1], torch.randn(2, 4, 8, generator=torch.Generator().manual_seed(6)))
    b = d.randn(2, 2, 4, 8, rng=torch.Generator().manual_seed(5))
    assert torch.equal(b, torch.randn(2, 2, 4, 8, generator=torch.Generator().manual_seed(5)))
    with pytest.raises(AssertionError):
        d.randn(3, 2, 4, 8, rng=gens)
    with pytest.raises(ValueError):
        d.randn(2, 2, rng="seed")


def test_config_roundtrip_and_validation():
    from r2dm_amd.option import Config

    c = Config(**synthetic_ckpt(resolution=GOLDEN_RES)["cfg"])
    assert c.data.resolution == GOLDEN_RES and c.model.coords_encoding == "fourier_features"
    assert c.data.min_depth == 1.45 and c.diffusion.timestep_type == "continuous"
    with pytest.raises(TypeError):
        Config(model={"no_such_field": 1})
    with pytest.raises(ValueError):
        Config(diffusion={"prediction_type": "score"})


def test_lidar_utility_cpu_members(golden):
    """The non-fused LiDARUtility members are device-agnostic torch expressions (not the hot path)."""
    from r2dm_amd.lidar import LiDARUtility

    g = golden("lidar")
    lu = LiDARUtility(GOLDEN_RES, "log_depth", 1.45, 80.0, ray_angles=g["ray_angles"])
    s = lu.denormalize(g["x"])
    depth = lu.revert_depth(s[:, [0]])
    y = torch.cat([depth, lu.to_xyz(depth), s[:, [1]]], 1)
    assert (y - g["y"]).abs().max() < 1e-5
    back = lu.convert_depth(depth)
    m = lu.get_mask(depth)
    assert ((back - s[:, [0]]) * m).abs().max() < 1e-5


def test_non_default_schedules_match_reference(golden):
    """S4 / S5 variants (continuous_time.py:18-58, discrete_time.py:22-48): the other log-SNR schedules and the cosine /
    sigmoid beta tables, bit-identical to what the reference computes (golden 'variants')."""
    from r2dm_amd import diffusion as D

    class Stub(torch.nn.Module):
        resolution, in_channels = GOLDEN_RES, 2

    g = golden("variants")
    t = g["t"]
    cases = {
        "linear": D.ContinuousTimeGaussianDiffusion(Stub(), noise_schedule="linear"),
        "cosine_shifted": D.ContinuousTimeGaussianDiffusion(Stub(), noise_schedule="cosine_shifted", image_d=64.0, noise_d_low=32.0),
        "cosine_interpolated": D.ContinuousTimeGaussianDiffusion(Stub(), noise_schedule="cosine_interpolated", image_d=64.0,
                                                                 noise_d_low=32.0, noise_d_high=256.0),
    }
    for name, d in cases.items():
        cond, coef, _ = d._coefficients(t[:-1], t[1:], "ddpm", 0.0)  # row by row on 1-element tensors, like the reference
        assert torch.equal(cond, g[f"lam_{name}"][:-1]), name
        lam_s = g[f"lam_{name}"][1:]
        assert torch.equal(coef[:, 2], torch.cat([lam_s[i:i + 1].sigmoid().sqrt() for i in range(len(lam_s))])), name
    with pytest.raises(AssertionError):
        D.ContinuousTimeGaussianDiffusion(Stub(), noise_schedule="cosine_shifted")  # needs image_d / noise_d_low
    for name in ("cosine", "sigmoid"):
        for T in (50, 1000):
            beta, ab, abp, snr = D.discrete_tables(T, name)
            assert torch.equal(beta, g[f"beta_{name}_{T}"]) and torch.equal(ab, g[f"alpha_bar_{name}_{T}"]), (name, T)
            assert abp[0] == 1 and torch.equal(abp[1:], ab[:-1])


def test_coordinate_encodings_match_reference(golden):
    """(f).4: spherical-harmonics / polar-coordinate encodings (encoding.py:80-117, efficient_unet.py:220-226) as the
    host-precomputed constant `__cenc`, against the reference's own evaluation (golden 'variants')."""
    from r2dm_amd import encodings as E
    from r2dm_amd.unet import EfficientUNet

    g = golden("variants")
    for enc, ch in (("spherical_harmonics", 25), ("polar_coordinates", 2), ("fourier_features", 22), (None, 0)):
        assert E.coord_channels(enc, GOLDEN_RES) == ch
    net = EfficientUNet(in_channels=2, resolution=GOLDEN_RES, base_channels=64, coords_encoding="spherical_harmonics")
    assert net.in_conv.weight.shape[1] == 27 and "coords_encoding.freqs" not in net.state_dict()
    import r2dm_amd

    for enc in ("spherical_harmonics", "polar_coordinates"):  # (the golden was taken with the checkpoint's HDL-64E ray angles)
        ddpm, _, _ = r2dm_amd.setup_model(synthetic_ckpt(resolution=GOLDEN_RES, coords_encoding=enc), device="cpu", show_info=False)
        c = E.coords_constant(enc, ddpm.model.coords)
        assert c.shape == g[f"cenc_{enc}"].shape and (c - g[f"cenc_{enc}"]).abs().max() < 3e-7, enc
    with pytest.raises(ValueError):
        E.coord_channels("cubemap", GOLDEN_RES)


def test_range_guard_fallback_host_logic():
    """The host side of the fp16 range guard's fallback (r2dm_amd/diffusion.py::_early_range_check, EfficientUNet.strict_range): with
    strict_range the early check only runs for loops longer than 8 steps and raises; otherwise it runs for every step count and
    reports a fallback, through a torch.compile wrapper (`_orig_mod`) too.  (The kernels' side: tests/test_hip_range.py, -m gpu.)"""
    import r2dm_amd
    from r2dm_amd import diffusion
    from r2dm_amd._lib import R2DMError, R2DMRangeError

    class Fake:
        def __init__(self, strict, trips):
            self.strict_range, self.trips, self.calls = strict, trips, 0

        def check_range_or_fall_back(self):
            self.calls += 1
            if self.trips and self.strict_range:
                raise R2DMRangeError("an input of the fp16-operand convolution path may be outside the fp16 range")
            return self.trips

    class Wrapper:  # what torch.compile returns
        def __init__(self, m):
            self._orig_mod = m

    assert issubclass(R2DMRangeError, R2DMError)
    m = Fake(strict=False, trips=True)
    assert diffusion._early_range_check(m, 2) is True and diffusion._early_range_check(Wrapper(m), 256) is True and m.calls == 2
    m = Fake(strict=False, trips=False)
    assert diffusion._early_range_check(m, 2) is False and m.calls == 1
    m = Fake(strict=True, trips=True)
    assert diffusion._early_range_check(m, 8) is False and m.calls == 0  # short loops: the deferred check at the loop's end reports it
    with pytest.raises(R2DMRangeError):
        diffusion._early_range_check(m, 9)
    assert diffusion._early_range_check(object(), 100) is False  # any other denoiser: nothing to check
    ddpm, _, _ = r2dm_amd.setup_model(synthetic_ckpt(resolution=GOLDEN_RES), device="cpu", show_info=False, strict_range=True)
    assert ddpm.model.strict_range is True and ddpm.model.range_fallbacks == 0
    ddpm, _, _ = r2dm_amd.setup_model(synthetic_ckpt(resolution=GOLDEN_RES), device="cpu", show_info=False)
    assert ddpm.model.strict_range is False


def test_replay_recovers_from_a_late_range_guard_trip_host_logic():
    """VERDICT round 4, missing #2 / ADVICE (medium): a guard trip after step 0 used to raise at the loop's end with strict_range off.
    diffusion._Replay on the host, with a scalar "denoiser" whose guard trips at a chosen call: the loop returns to the last checked
    x, replays the steps since on the RECORDED noise and ends on exactly the value of a run whose denoiser was "wide" from the start;
    checks are due after step 0 (loops > 8 steps), every _CHECK_EVERY steps and at the end; strict_range raises instead."""
    from r2dm_amd import diffusion
    from r2dm_amd._lib import R2DMRangeError, R2DMRangeFallback

    assert issubclass(R2DMRangeFallback, R2DMRangeError)

    class Fake:
        def __init__(self, trip_at_call, strict=False):
            self.strict_range, self.trip_at, self.calls, self.wide, self.tripped, self.checks = strict, trip_at_call, 0, False, False, 0

        def __call__(self, x):
            self.calls += 1
            if not self.wide and self.calls >= self.trip_at:
                self.tripped = True  # (from here on the narrow path returns garbage)
            return x * 0.5 + (0.0 if self.wide or not self.tripped else 1e3)

        def check_range_or_fall_back(self):
            self.checks += 1
            if self.tripped and not self.wide:
                if self.strict_range:
                    raise R2DMRangeError("may be outside the fp16 range")
                self.wide, self.tripped = True, False
                return True
            return False

    def run(model, n):
        rp = diffusion._Replay(model, n)
        x, draws = 1.0, iter(range(1000))
        rp.start(x)
        i, outs = 0, [x]
        while i < n:
            nz = rp.noise_for(i, lambda: float(next(draws)))  # (a replayed step must NOT draw again)
            nxt, x = rp.after_step(i, model(x) + 1e-3 * nz)
            del outs[nxt + 1:]
            if nxt > i:
                outs.append(x)
            i = nxt
        return x, outs, rp

    ref, ref_outs, _ = run(Fake(10 ** 9), 100)
    for trip in (1, 2, 33, 34, 70, 100):
        m = Fake(trip)
        x, outs, rp = run(m, 100)
        assert x == ref and outs == ref_outs and rp.replays == 1 and m.wide, trip
        assert m.calls <= 100 + diffusion._CHECK_EVERY, (trip, m.calls)  # at most one window is run twice
    m = Fake(10 ** 9)
    run(m, 100)
    assert m.checks == 1 + 3 + 1  # after step 0, after steps 32 / 64 / 96, at the end
    m = Fake(10 ** 9)
    run(m, 8)
    assert m.checks == 1  # short loops: the end only
    with pytest.raises(R2DMRangeError):
        run(Fake(1, strict=True), 100)
    x, _, rp = run(object(), 5) if False else (None, None, diffusion._Replay(object(), 5))
    assert rp.check is None and rp.due(0) is False and rp.due(4) is False  # any other denoiser: nothing to check, nothing kept


def test_blob_layout_fingerprint():
    """the f16x2 packing of a layer depends on the tile conv_f16x2_pick_co_tile chooses -- from max_batch, the CU
    count, experiment switches -- and the byte count cannot tell two such layouts apart.  r2dm_blob_layout_hash fingerprints the plan;
    a blob offered with another plan's fingerprint is refused before anything is bound (host-only: no GPU needed)."""
    import r2dm_amd
    from r2dm_amd._lib import R2DMError

    ck = synthetic_ckpt()
    m2, _, _ = r2dm_amd.setup_model(ck, device="cpu", show_info=False, max_batch=2)
    m2b, _, _ = r2dm_amd.setup_model(ck, device="cpu", show_info=False, max_batch=2)
    m8, _, _ = r2dm_amd.setup_model(ck, device="cpu", show_info=False, max_batch=8)
    h2, h8 = m2.model.packed_layout_hash(), m8.model.packed_layout_hash()
    assert h2 == m2b.model.packed_layout_hash() and h2 != h8 and h2 != 0
    with pytest.raises(R2DMError, match="layout"):
        m8.model.adopt_packed_weights(torch.empty(m8.model.packed_weight_bytes(), dtype=torch.uint8), layout_hash=h2)


def test_no_packed_fp32_with_scalar_operands_in_kernels_that_share_cus(tmp_path):
    """Round 5 (profiles/r05_coresidency.txt): a packed-fp32 VALU instruction (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) whose source is an SGPR pair read wrong
    values in lanes 48-63 whenever its wave shared the CU with an LDS-holding workgroup of ANOTHER PROCESS -- the root of the "exact alone, wrong next to a second
    process" family of rounds 2, 3 and 5.  Invariant checked on the built code objects (no GPU needed): outside conv_f16x2_kernel, whose blocks own their CUs, and
    the kernels compiled without SGPR-operand forms, NO kernel of the library contains such an instruction (r2dm_amd/csrc/build.sh compiles the CU-sharing sources
    that would get them -- attention, in_conv / out_conv, FIR, posterior -- without packed-fp32 instruction selection)."""
    import shutil
    import subprocess

    llvm = "/opt/rocm/lib/llvm/bin"
    lib = os.path.join(ROOT, "r2dm_amd", "libr2dm_hip.so")
    if not os.path.exists(os.path.join(llvm, "llvm-objdump")):
        pytest.skip("llvm-objdump of the ROCm toolchain not present")
    if not os.path.exists(lib):
        pytest.skip("libr2dm_hip.so not built")
    shutil.copy(lib, tmp_path / "lib.so")
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp_path, check=True, capture_output=True)
    offenders, kernels = {}, 0
    for f in sorted(os.listdir(tmp_path)):
        if "gfx950" not in f:
            continue
        dis = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", f], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        name = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                name = m.group(1)
                kernels += 1
                continue
            if name and re.search(r"\bv_pk_\w+_f32\b", line) and re.search(r"\bs\[\d+:\d+\]", line.split("//")[0]):
                if "conv_f16x2_kernel" in name:  # (owns its CUs: test_conv_f16x2_blocks_own_their_cu)
                    continue
                offenders[name] = offenders.get(name, 0) + 1
    assert kernels > 50, kernels
    assert not offenders, offenders


def test_conv_f16x2_blocks_own_their_cu(tmp_path):
    """Round 5 (profiles/r05_coresidency.txt): next to ANOTHER PROCESS's waves on its CU the one-plane 32-channel tile of
    conv_f16x2.hip ended most 600-forward runs in a GPU memory fault; with the CU to itself never.  The invariant since then: every
    conv_f16x2_kernel instantiation allocates all 256 vector registers (two waves per SIMD = the whole file) and every launch requests
    >= 156 KiB of LDS -- nothing else fits on a CU that runs one of its blocks.  Checked here on the BUILT code objects (their metadata
    notes), so a compiler or source change that shrinks the allocation fails on the CPU suite
    (/root/reference/sample_and_save.py:37-46: results must not depend on what else the GPU runs)."""
    import shutil
    import subprocess

    llvm = "/opt/rocm/lib/llvm/bin"
    lib = os.path.join(ROOT, "r2dm_amd", "libr2dm_hip.so")
    if not (os.path.exists(os.path.join(llvm, "llvm-objdump")) and os.path.exists(os.path.join(llvm, "llvm-readelf"))):
        pytest.skip("llvm-objdump / llvm-readelf of the ROCm toolchain not present")
    if not os.path.exists(lib):
        pytest.skip("libr2dm_hip.so not built")
    shutil.copy(lib, tmp_path / "lib.so")
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp_path, check=True, capture_output=True)
    counts = []
    for f in sorted(os.listdir(tmp_path)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", f], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        if "conv_f16x2_kernel" not in notes:
            continue
        # one metadata record per kernel: ".name: <mangled>" ... ".vgpr_count: N" (keys sorted alphabetically inside a record)
        for rec in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", rec).group(1)
            if "4r2dm17conv_f16x2_kernel" in name:  # (not pack_conv_f16x2_kernel)
                counts.append((name, int(re.search(r"\.vgpr_count:\s+(\d+)", rec).group(1)), int(re.search(r"\.agpr_count:\s+(\d+)", "- .agpr_count:" + rec).group(1))))
    assert len(counts) >= 30, "conv_f16x2_kernel instantiations not found in the library's code objects"
    assert all(v + a == 256 for _, v, a in counts), [c for c in counts if c[1] + c[2] != 256][:4]
    src = open(os.path.join(ROOT, "r2dm_amd", "csrc", "conv_f16x2.hip")).read()
    assert re.search(r"LDS_TOTAL >= 156 \* 1024 \|\| lds_exact \? GEO::LDS_TOTAL : 156 \* 1024", src), "the launcher's LDS padding"


def test_bench_gpus_flag_is_checked_before_anything_runs():
    """VERDICT round 5, item 2 (no GPU needed): `--gpus N` must agree with the launcher's WORLD_SIZE, and N ranks under RCCL need N devices --
    both fail in seconds, non-zero, with a message that names the numbers (here the device count is 0: no GPU in the CPU tier)."""
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "R2DM_DIST_BACKEND")}
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and f"device_count() = {torch.cuda.device_count()}" in r.stderr and "--gpus 2" in r.stderr
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=dict(env, WORLD_SIZE="4", R2DM_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def test_schedule_on_option_and_replay_recording():
    """setup_model(schedule_on=...) is validated and switchable (the device evaluation itself: tests/test_hip_configs.py); and the sampling
    loop's replay state is kept only for models whose range guard can fall back (not for 'fp32-bf16x3')."""
    import r2dm_amd
    from r2dm_amd.diffusion import _Replay

    ddpm, _, _ = r2dm_amd.setup_model(synthetic_ckpt(), device="cpu", show_info=False)
    assert ddpm.schedule_on == "host"
    ddpm.set_schedule_on("device")
    assert ddpm.schedule_on == "device" and not ddpm._on_device("cpu")  # (CPU tensors: the host path, whatever the option says)
    with pytest.raises(ValueError):
        ddpm.set_schedule_on("gpu")
    with pytest.raises(ValueError):
        r2dm_amd.setup_model(synthetic_ckpt(), device="cpu", show_info=False, schedule_on="nowhere")
    x = torch.zeros(1)
    for prec, rec in (("fp32", True), ("fp16", True), ("fp32-bf16x3", False)):
        ddpm.model.set_precision(prec)
        rp = _Replay(ddpm.model, 64)
        rp.start(x)
        z = rp.noise_for(0, lambda: torch.ones(1))
        assert rp.record is rec and (rp.ck_x is x) is rec and len(rp.noise) == (1 if rec else 0) and z.item() == 1.0


def test_out_conv_keeps_its_loads_in_flight(tmp_path):
    """Round 6: out_conv (conv_direct_rows_kernel<2, false>: one wave per SIMD, latency hidden by the loads of four unrolled input channels in flight) once
    compiled to 72 instead of 122 registers after its loop body was shared with the fp16-input variant -- fewer loads in flight, 53 -> 117 us per step, found
    only in a profile.  The allocation is visible in the code object's metadata: checked here, on the CPU."""
    import shutil
    import subprocess

    llvm = "/opt/rocm/lib/llvm/bin"
    lib = os.path.join(ROOT, "r2dm_amd", "libr2dm_hip.so")
    if not (os.path.exists(os.path.join(llvm, "llvm-objdump")) and os.path.exists(os.path.join(llvm, "llvm-readelf"))):
        pytest.skip("llvm-objdump / llvm-readelf of the ROCm toolchain not present")
    if not os.path.exists(lib):
        pytest.skip("libr2dm_hip.so not built")
    shutil.copy(lib, tmp_path / "lib.so")
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp_path, check=True, capture_output=True)
    found = None
    for f in sorted(os.listdir(tmp_path)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", f], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        for rec in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", rec).group(1)
            if "conv_direct_rows_kernelILi2ELb0E" in name:
                found = int(re.search(r"\.vgpr_count:\s+(\d+)", rec).group(1))
    assert found is not None, "conv_direct_rows_kernel<2, false> not found in the library's code objects"
    assert found >= 100, f"out_conv allocates {found} registers: its loads are no longer four channels deep"
