#!/usr/bin/env python3
"""GPU-box fuzz of the SAMPLER surface (continuous time): random objective / mode / eta / step count / batch vs max_batch / RNG form /
schedule at 16x128 against the oracle's sampler (float32 sampler arithmetic as the reference's, float64 denoiser on the device) driven by
the same generators.  The first steps from t = 1 are ill-conditioned, so the statement is the 99th percentile
and the rms of |hip - oracle|, finiteness, the reference's exceptions for bad arguments, and return_all's shape."""
import os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r2dm_amd
from r2dm_amd import synthetic
from oracle import r2dm_oracle as O
dev = torch.device("cuda", 0)
rnd = random.Random(int(os.environ.get("SEED", "1")))
N = int(os.environ.get("CASES", "40"))
RES = (16, 128)
fails = 0; worst_q99 = 0.0
for case in range(N):
    obj = rnd.choice(["eps", "eps", "v", "x_0"]); mode = rnd.choice(["ddpm", "ddim"]); eta = rnd.choice([0.0, 0.0, 0.3, 1.0]) if mode == "ddim" else 0.0
    sched = rnd.choice(["cosine", "cosine", "linear"]); S = rnd.randint(1, 6); maxb = rnd.choice([1, 2, 4, 8]); B = rnd.randint(1, 2 * maxb)
    rngk = rnd.choice(["list", "list", "single", "none"]); ra = rnd.random() < 0.3; prec = rnd.choice(["fp32", "fp32", "fp32-bf16x3"])
    tag = f"case {case}: {obj} {mode} eta {eta} {sched} S {S} batch {B} (max_batch {maxb}) rng {rngk} return_all {ra} {prec}"
    ck = synthetic.synthetic_checkpoint(seed=0, resolution=RES, prediction_type=obj, noise_schedule=sched)
    ddpm, _, _ = r2dm_amd.setup_model(ck, device=dev, show_info=False, max_batch=maxb, precision=prec)
    mk = {"list": lambda: r2dm_amd.setup_rng(list(range(100, 100 + B)), dev), "single": lambda: torch.Generator(device=dev).manual_seed(7), "none": lambda: None}[rngk]
    if rngk == "none": torch.manual_seed(5)
    got = ddpm.sample(batch_size=B, num_steps=S, progress=False, rng=mk(), return_all=ra, mode=mode, ddim_eta=eta)
    sd = {k: v.double().to(dev) for k, v in O.strip_prefix(ck["ema_weights"]).items()}
    cfg = O.UNetConfig(resolution=RES)
    net = lambda x, c: O.unet_forward(sd, cfg, x.double(), c.double()).float()
    if rngk == "none": torch.manual_seed(5)
    want = O.sample_continuous(net, (B, 2, *RES), S, rng=mk(), return_all=ra, mode=mode, ddim_eta=eta, objective=obj, device=dev,
                               log_snr=O.log_snr_cosine if sched == "cosine" else O.log_snr_linear)
    shape_ok = tuple(got.shape) == tuple(want.shape) == ((S + 1, B, 2, *RES) if ra else (B, 2, *RES))
    d = (got - want).abs().flatten().double()
    q99 = torch.quantile(d[:: max(1, d.numel() // 200000)], 0.99).item(); rms = d.pow(2).mean().sqrt().item()
    # DDIM with eta = 1: the reference's c_2 = sqrt(1 - alpha_s^2 - c_1^2) (continuous_time.py:224) cancels: on the LAST step of a 4-step cosine
    # schedule its float32 value is 2.3e-4 where float64 gives 7.4e-7 (the argument, 5e-8, is the rounding of 1 - ...), on the first 1.269e-3 vs
    # 1.233e-3.  So c_2 there is rounding noise of ~1e-4, and two evaluations of the scalars (the product: host CPU, bit-identical to the
    # reference's CPU run, tests/test_host.py; this oracle: the device's libm) differ by that much on the sample.  Looser bar, stated here.
    loose = mode == "ddim" and eta >= 1.0
    ok = shape_ok and bool(torch.isfinite(got).all()) and q99 < (3e-4 if loose else 2e-5) and rms < (3e-4 if loose else 2e-4)
    fails += (not ok); worst_q99 = max(worst_q99, q99)
    print(tag, f"-> q99 {q99:.2e} rms {rms:.2e} max {d.max().item():.2e} {'OK' if ok else 'FAIL'}", flush=True)
# the reference's argument errors (continuous_time.py:215,231; base.py:82)
ddpm, _, _ = r2dm_amd.setup_model(synthetic.synthetic_checkpoint(seed=0, resolution=RES), device=dev, show_info=False, max_batch=2)
for kw, exc in ((dict(mode="heun"), ValueError), (dict(rng=r2dm_amd.setup_rng([1], dev)), AssertionError)):
    try:
        ddpm.sample(batch_size=2, num_steps=2, progress=False, **kw); print("bad arguments", kw.keys(), "did NOT raise: FAIL"); fails += 1
    except exc:
        print("bad arguments", list(kw), "->", exc.__name__, "OK")
print(f"{N} cases, {fails} failures, worst q99 {worst_q99:.2e}")
