"""Checkpoint pre-flight for the default ("fp32": 22-bit split fp16 operands) mode:

    python -m r2dm_amd.check <checkpoint.pth | config name> [--steps 256] [--batch 1] [--mode ddpm] [--device cuda:0] [--seed 0]

The default mode's matrix products read their operands through fp16 (|operand| < 65504), behind a data-driven guard; a checkpoint whose
activations leave that range still samples -- the model switches itself to the wide-range split "fp32-bf16x3" -- but about 1.7x slower
(DESIGN.md section 2).  Whether a TRAINED checkpoint (/root/reference/hubconf.py:17-37: r2dm-h-kitti360-300k and friends; no network in
the build container, so none has been through the engine) stays inside the guard at every log-SNR of its sampler is what this tool
answers, in one command, for whoever has the file:

  * it runs the real sampling loop (``ddpm.sample``'s steps: the checkpoint's own schedule, per-sample generators seeded ``--seed ...``)
    and reads the guard after EVERY step (`EfficientUNet.range_report`: one bound per guarded layer);
  * a step whose guard trips is repeated on the wide-range split so that the trajectory stays the one the sampler would follow;
  * it prints, per guarded layer, the largest bound / 65504 over all steps (the headroom used) and the log-SNR where it peaked, the steps
    that would have fallen back, and exits 1 if any did (0: the checkpoint runs the fast path at every step).
"""
from __future__ import annotations

import argparse
import sys
import warnings

import torch


def preflight(ckpt, num_steps: int = 256, batch: int = 1, mode: str = "ddpm", ddim_eta: float = 0.0, device="cuda:0", seed: int = 0,
              ema: bool = True):
    """-> dict(sites=[(name, worst bound, step, log-SNR)], trips=[(step, log-SNR, bound, site)], steps=num_steps).  `ckpt`: path or dict."""
    from . import _lib, inference

    dev = torch.device(device)
    ddpm, _, _ = inference.setup_model(ckpt, device=dev, ema=ema, show_info=False, max_batch=batch, precision="fp32", strict_range=True)
    net = ddpm.model
    rng = inference.setup_rng(list(range(seed, seed + batch)), dev)
    x = ddpm.randn(batch, *ddpm.sampling_shape, rng=rng, device=dev)
    discrete = not hasattr(ddpm, "_table_rows")
    if discrete:
        ts = list(range(num_steps - 1, -1, -1))
    else:
        steps = torch.linspace(1.0, 0.0, num_steps + 1)
    worst, trips = {}, []
    net._defer_range_check = True  # (the forwards do not check themselves: this loop reads the guard after every step)
    with torch.inference_mode():
        for i in range(num_steps):
            if discrete:
                t = torch.full((batch,), ts[i], device=dev, dtype=torch.long)
                cond_val = float(ts[i])
            else:
                t, s = steps[i][None].repeat_interleave(batch), steps[i + 1][None].repeat_interleave(batch)
                cond_val = float(ddpm.get_network_condition(steps[i:i + 1])[0])
            states = [g.get_state() for g in rng]

            def one_step():
                if discrete:
                    return ddpm.p_step(x, t, rng=rng, mode=mode)
                return ddpm.p_step(x, t, s, rng=rng, mode=mode, ddim_eta=ddim_eta)

            x_next = one_step()
            tripped = False
            try:
                net.check_range()
            except _lib.R2DMRangeError:
                tripped = True
            rep = net.range_report()
            for name, b in rep:
                if name not in worst or b > worst[name][0]:
                    worst[name] = (b, i, cond_val)
            if tripped:
                name, b = max(rep, key=lambda e: e[1]) if rep else ("?", float("inf"))
                trips.append((i, cond_val, b, name))
                for g, st in zip(rng, states):  # the step again, on the wide-range split and the same draws: the sampler's own trajectory
                    g.set_state(st)
                net.set_precision("fp32-bf16x3")
                x_next = one_step()
                net.set_precision("fp32")
            x = x_next
    if not torch.isfinite(x).all():
        warnings.warn("the final sample is not finite", RuntimeWarning)
    sites = sorted(((n, b, i, c) for n, (b, i, c) in worst.items()), key=lambda e: -e[1])
    return {"sites": sites, "trips": trips, "steps": num_steps, "condition": "timestep" if discrete else "log-SNR"}


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m r2dm_amd.check", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("ckpt", help="checkpoint file (the dict train.py saves) -- or 'synthetic' for the deterministic synthetic checkpoint")
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--mode", choices=["ddpm", "ddim"], default="ddpm")
    ap.add_argument("--ddim-eta", type=float, default=0.0)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-ema", action="store_true")
    ap.add_argument("--top", type=int, default=0, help="print only the N layers closest to the limit (default: all)")
    a = ap.parse_args(argv)
    ckpt = a.ckpt
    if ckpt == "synthetic":
        from . import synthetic

        ckpt = synthetic.synthetic_checkpoint(seed=0)
    r = preflight(ckpt, a.steps, a.batch, a.mode, a.ddim_eta, a.device, a.seed, ema=not a.no_ema)
    print(f"# r2dm_amd.check: {a.steps}-step {a.mode.upper()} sampler, batch {a.batch}, seeds {a.seed}..{a.seed + a.batch - 1}; guarded layers: {len(r['sites'])}")
    print(f"# {'bound / 65504':>14s}  {'bound':>10s}  {'step':>5s}  {r['condition']:>9s}  layer: what the guard bounds")
    for name, b, i, c in (r["sites"][: a.top] if a.top else r["sites"]):
        print(f"  {b / 65504.0:14.3e}  {b:10.4g}  {i:5d}  {c:9.3f}  {name}")
    if r["trips"]:
        print(f"# {len(r['trips'])} of {a.steps} steps would fall back to the wide-range split 'fp32-bf16x3' (the model switches for good at the first):")
        for i, c, b, name in r["trips"][:16]:
            print(f"  step {i:4d}  {r['condition']} {c:8.3f}  bound {b:.4g}  at {name}")
        print("# -> this checkpoint samples on 'fp32-bf16x3' (about 1.7x slower); pass precision='fp32-bf16x3' to setup_model to start there")
        return 1
    head = r["sites"][0][1] / 65504.0 if r["sites"] else 0.0
    print(f"# no step trips the guard: the checkpoint runs the default mode at every step (largest bound = {head:.3g} of the fp16 range)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
