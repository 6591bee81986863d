#!/usr/bin/env python3
"""Benchmark: range-images/sec of the reverse-process sampler (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {1,2,4}]

A "step" (driver vocabulary) is ONE reverse step over the whole per-GPU batch: one U-Net forward
(r2dm_unet_forward) + RNG draw + fused posterior update.  Workloads (BASELINE.json `configs`):

    --config 1 (default, the metric's configuration)  64x1024x2, batch 8 per GPU, DDPM, 256-step sampler
    --config 2                                        64x1024x2, batch 32 per GPU, DDIM (eta 0), 32-step sampler
    --config 4                                        128x2048x2, batch 2 per GPU (16 over 8 GPUs), DDPM, 256-step sampler

Throughput in images/s is quoted for the configuration's full sampler, value = n_gpus * batch / (S * seconds_per_step);
timing K steps inside ONE sample() call of K steps (default K = 16 so the default run finishes in seconds; --steps 256
times the full sampler) is exact for this metric because the per-step work does not depend on the step index.  Before the
warm-up steps the sampler runs untimed until the board's shader clock and power are stationary (`config.clock_prewarm`):
the convolutions run into the board power limit, and a short timed call on a cool board reads 5-7 % faster than the
sustained 256-step rate it is scaled to (round 2: 4.75 vs 4.53 images/s in one job).
For N > 1 launch under torch.distributed.run (one rank per GPU): rank 0 packs the weights, the packed blob is
broadcast over RCCL/xGMI, every rank samples its own seeds; no collective in the step loop.

Two baselines ride on the same JSON line at N = 1 (rank 0): `cpu_baseline` = the oracle (torch CPU ops) on the host
cores, and `torch_rocm_baseline` = the same oracle on the MI355X through stock PyTorch-ROCm (MIOpen / rocBLAS) -- the
stand-in for north_star's "reference single-GPU PyTorch sampler" (the reference's own files never travel to the GPU
box; the oracle is pinned to it by tests/golden); `vs_torch_rocm_baseline` = value / that (`vs_baseline` is null: BASELINE.md
holds no published number for the metric).  `exact_split_baseline` = the same timed
call with `--precision fp32-bf16x3` (exact 24-bit operands, six bf16 products): the headline's arithmetic is the 22-bit
fp16 split (three products, the residual x residual term dropped; fp32-class by measurement), and the line shows what the
fully exact operand split costs beside it.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# name -> (resolution, per-GPU batch, sampler steps, mode, algorithmic FLOP per image-step [SURVEY.md section 8(d), 2*MAC])
CONFIGS = {
    1: dict(res=(64, 1024), batch=8, sampler_steps=256, mode="ddpm", flop=234.52e9,
            metric="range-images/sec (64x1024, 256-step DDPM)",
            workload="BASELINE configs[1]: 64x1024x2 range/reflectance, 256-step DDPM, batch 8 per GPU"),
    2: dict(res=(64, 1024), batch=32, sampler_steps=32, mode="ddim", flop=234.52e9,
            metric="range-images/sec (64x1024, 32-step DDIM)",
            workload="BASELINE configs[2]: 64x1024x2 range/reflectance, 32-step DDIM (eta 0), batch 32 per GPU"),
    4: dict(res=(128, 2048), batch=2, sampler_steps=256, mode="ddpm", flop=977.92e9,
            metric="range-images/sec (128x2048, 256-step DDPM)",
            workload="BASELINE configs[4] geometry: 128x2048x2, 256-step DDPM, batch 2 per GPU (16 over 8 GPUs)"),
}
PEAK_FP32 = 157.3e12            # MI355X fp32 vector == fp32-input MFMA peak (MI355X_MICROARCH.md)
PEAK_BF16 = 2500e12             # dense 16-bit (bf16 / fp16) MFMA peak at 2.4 GHz (MI355X_MICROARCH.md)
# The convolutions (97 % of the FLOPs) run on the 16-bit matrix pipe with every fp32 operand split into 16-bit pieces, and
# the hardware ceiling for their ALGORITHMIC fp32 FLOPs is the dense 16-bit peak / (matrix products per fp32 product):
#   precision fp32 (default)   fp16 + 2^11-scaled fp16 residual = 22 bits, 3 products (lo x lo dropped)  -> 2500 / 3
#   precision fp32-bf16x3      three bf16 pieces = 24 bits (exact), 6 products                           -> 2500 / 6
#   precision fp16             the fp16 piece alone, 1 product (reduced-precision bulk mode)             -> 2500 / 1


def oracle_net(ck, res, device):
    from oracle import r2dm_oracle as O

    sd = {k: v.to(device) for k, v in O.strip_prefix(ck["ema_weights"]).items()}
    cfg = O.UNetConfig(resolution=res)
    return O, (lambda x, c: O.unet_forward(sd, cfg, x, c))


def cpu_baseline(ck, cf):
    """The oracle (a torch-op restatement of the reference, pinned to it by tests/golden) on the host cores:
    BASELINE configs[0] shape -- batch 1, same resolution and sampler -- bounded to a few steps (~10-20 s of CPU work).
    Thread count: oneDNN's fp32 convolutions at batch 1 do not scale past ~16 threads on this host class
    (measured on the 256-core GPU box: 0.26 s/forward at 16 threads, 0.63 s at 32, 1.3 s at 64, 5.1 s at 128),
    so the baseline uses min(16, cores) and reports that as `cores`."""
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    O, net = oracle_net(ck, cf["res"], "cpu")
    rng = [torch.Generator().manual_seed(0)]
    n = 8 if cf["res"] == (64, 1024) else 4  # (64x1024: BASELINE configs[0] literally -- batch 1, 8 DDPM steps)
    with torch.inference_mode():
        O.sample_continuous(net, (1, 2, *cf["res"]), 1, rng=rng, mode=cf["mode"])  # warm-up
        t0 = time.perf_counter()
        O.sample_continuous(net, (1, 2, *cf["res"]), n, rng=rng, mode=cf["mode"])
        dt = time.perf_counter() - t0
    return {"value": 1.0 / (dt / n * cf["sampler_steps"]), "unit": "images/s", "cores": cores, "kind": "port",
            "configs0_seconds": dt if n == 8 and cf["mode"] == "ddpm" else None,  # wall time of BASELINE configs[0] itself (8-step DDPM, batch 1)
            "sample": f"oracle (torch CPU ops, fp32) sample(batch=1, {n} {cf['mode'].upper()} steps) at {cf['res'][0]}x{cf['res'][1]} on "
                      f"{cores} threads of {os.cpu_count()} host cores, {dt / n:.3f} s/step, scaled to the {cf['sampler_steps']}-step sampler"}


def torch_rocm_baseline(ck, cf, B, dev, compiled=True, compile_budget_s=240):
    """The same oracle evaluated by stock PyTorch-ROCm on the MI355X (MIOpen convolutions, rocBLAS attention), fp32,
    same batch: what running the reference's PyTorch sampler on this GPU costs.  One warm-up step (MIOpen kernel
    selection), then a bounded number of steps timed with a device synchronize on both sides."""
    O, net = oracle_net(ck, cf["res"], dev)
    rng = [torch.Generator(device=dev).manual_seed(i) for i in range(B)]
    shape = (B, 2, *cf["res"])
    n = 3
    with torch.inference_mode():
        O.sample_continuous(net, shape, 1, rng=rng, mode=cf["mode"], device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        O.sample_continuous(net, shape, n, rng=rng, mode=cf["mode"], device=dev)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out = {"value": B / (dt / n * cf["sampler_steps"]), "unit": "images/s", "ms_per_step": dt / n * 1e3,
           "kind": "oracle via PyTorch-ROCm (torch %s: MIOpen / rocBLAS, fp32, eager)" % torch.__version__,
           "sample": f"sample(batch={B}, {n} {cf['mode'].upper()} steps after 1 warm-up step), scaled to the {cf['sampler_steps']}-step sampler"}
    # ... and the way the reference's BULK sampler runs it: the denoiser under fp16 autocast (sample_and_save.py:70) -- the counterpart of
    # `--precision fp16` here, reported beside the fp32 figure (vs_torch_rocm_baseline uses the fp32 one: same arithmetic class as the headline)
    try:
        anet = lambda x, c: torch.autocast("cuda", dtype=torch.float16)(net)(x, c).float()
        with torch.inference_mode():
            O.sample_continuous(anet, shape, 1, rng=rng, mode=cf["mode"], device=dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            O.sample_continuous(anet, shape, n, rng=rng, mode=cf["mode"], device=dev)
            torch.cuda.synchronize()
            dta = time.perf_counter() - t0
        out["fp16_autocast"] = {"value": B / (dta / n * cf["sampler_steps"]), "unit": "images/s", "ms_per_step": dta / n * 1e3,
                                "kind": "the same oracle under torch.autocast(fp16), eager"}
    except Exception as e:  # (an autocast path MIOpen cannot serve must not take the line down)
        out["fp16_autocast"] = {"error": repr(e)[:200]}
    # ... and the form the reference's bulk script REALLY runs (sample_and_save.py:45,70): the denoiser through torch.compile (inductor)
    # under fp16 autocast.  Compilation is bounded (a watchdog ends it: the default bench run has minutes, not an hour).
    if compiled:
        import signal

        def _alarm(*_):
            raise TimeoutError("torch.compile did not finish inside the watchdog")

        old = signal.signal(signal.SIGALRM, _alarm)
        signal.alarm(int(compile_budget_s))
        try:
            t_c = time.perf_counter()
            cnet = torch.compile(net)
            canet = lambda x, c: torch.autocast("cuda", dtype=torch.float16)(cnet)(x, c).float()
            with torch.inference_mode():
                O.sample_continuous(canet, shape, 1, rng=rng, mode=cf["mode"], device=dev)
                torch.cuda.synchronize()
                compile_s = time.perf_counter() - t_c
                t0 = time.perf_counter()
                O.sample_continuous(canet, shape, n, rng=rng, mode=cf["mode"], device=dev)
                torch.cuda.synchronize()
                dtc = time.perf_counter() - t0
            out["compiled_fp16_autocast"] = {"value": B / (dtc / n * cf["sampler_steps"]), "unit": "images/s", "ms_per_step": dtc / n * 1e3, "compile_s": compile_s,
                                             "kind": "the same oracle through torch.compile (inductor) under torch.autocast(fp16): how /root/reference/sample_and_save.py:45,70 runs the denoiser"}
        except (Exception, TimeoutError) as e:  # (incl. the watchdog; a compile failure must not take the line down -- KeyboardInterrupt / SystemExit pass)
            out["compiled_fp16_autocast"] = {"error": repr(e)[:300]}
        finally:
            signal.alarm(0)
            signal.signal(signal.SIGALRM, old)
    return out


class BoardSampler:
    """Shader clock and board power of the device the timed region runs on, sampled from the amdgpu hwmon files by a
    thread (20 ms period).  The split-operand convolutions run into the 1400 W board power limit, so the sustained clock
    -- not the 2.4 GHz the nominal MFMA peak is quoted at -- sets what the matrix pipe can deliver."""

    def __init__(self, dev_index):
        import glob
        import threading
        self.dir = None
        pr = torch.cuda.get_device_properties(dev_index)
        want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}" if hasattr(pr, "pci_bus_id") else None
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            try:
                slot = [ln.strip().split("=")[1] for ln in open(os.path.dirname(os.path.dirname(d)) + "/uevent") if ln.startswith("PCI_SLOT_NAME=")]
            except OSError:
                slot = []
            if want and slot and slot[0].startswith(want) and os.path.exists(d + "/freq1_input"):
                self.dir = d
        self.samples, self._stop, self._th = [], False, threading.Thread(target=self._run, daemon=True)

    def _rd(self, name):
        try:
            with open(f"{self.dir}/{name}") as f:
                return int(f.read())
        except (OSError, ValueError):
            return None

    def _run(self):
        while not self._stop:
            self.samples.append((self._rd("freq1_input"), self._rd("power1_average") or self._rd("power1_input")))
            time.sleep(0.02)

    def __enter__(self):
        if self.dir:
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self.dir:
            self._th.join()

    def summary(self, all_samples=False):
        import statistics
        s = self.samples if all_samples else self.samples[len(self.samples) // 3:]  # steady part
        f = [a for a, _ in s if a]
        w = [b for _, b in s if b]
        if not f:
            return None
        cap = self._rd("power1_cap")
        return {"sclk_mhz": statistics.median(f) / 1e6, "board_w": statistics.median(w) / 1e6 if w else None,
                "power_cap_w": cap / 1e6 if cap else None, "samples": len(s), "source": self.dir}


def hipblaslt_reference(dev):
    """What the vendor library's bf16 GEMM (8192^3, torch.matmul -> hipBLASLt) sustains on this board, for scale: it runs
    into the same power limit.  ~1 s."""
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    for _ in range(20):
        a @ b
    torch.cuda.synchronize()
    n = 0
    with BoardSampler(dev.index) as bs:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 1.0:
            for _ in range(20):
                a @ b
            n += 20
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return {"tflops": 2 * 8192 ** 3 * n / dt / 1e12, "board": bs.summary(), "what": "torch.matmul bf16 8192x8192x8192 (hipBLASLt), 1 s"}


def pmc_traffic(what="bytes_per_launch"):
    """HBM bytes per conv launch from the committed PMC passes (scripts/summarize_profile.py); PMC counters cannot be
    collected from inside the timed process, so this is the figure of the last profiled run of this same command
    (what="source": the evidence set it was taken from)."""
    try:
        with open(os.path.join(ROOT, "profiles", "conv_traffic.json")) as f:
            d = json.load(f)
            return float(d["bytes_per_launch"]) if what == "bytes_per_launch" else str(d[what])
    except (OSError, KeyError, ValueError):
        return None


def launch_ranks(args):
    """`--gpus N` stands on its own (the reference gets its ranks from `accelerate launch`, /root/reference/sample_and_save.py:25-46):
    * WORLD_SIZE set (torch.distributed.run started us): it must equal --gpus, or the line would report a world the caller did not ask for;
    * WORLD_SIZE unset and N > 1: re-execute under `torch.distributed.run --nproc-per-node N` with the same arguments;
    * RCCL wants one device per rank: N > torch.cuda.device_count() fails here, in seconds, naming the device count (R2DM_DIST_BACKEND=gloo
      lets ranks share a GPU: the tests' two-ranks-on-one-GPU layout)."""
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus is None:
        args.gpus = int(env_world) if env_world else 1
    if args.gpus < 1:
        sys.exit(f"bench.py: --gpus {args.gpus}: need at least one GPU")
    backend = os.environ.get("R2DM_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if args.gpus > 1 and backend == "nccl" and args.gpus > ndev:
        sys.exit(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs under the RCCL backend (one rank per device), but this node has "
                 f"torch.cuda.device_count() = {ndev}; R2DM_DIST_BACKEND=gloo lets ranks share a device (tests only)")
    if env_world is not None:
        if int(env_world) != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={env_world}: start it as "
                     f"`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}` (or plain `python bench.py --gpus {args.gpus}`)")
        return
    if args.gpus == 1:
        return
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="number of GPUs = ranks of ONE node.  Under torch.distributed.run it must equal WORLD_SIZE; "
                    "without a launcher (WORLD_SIZE unset) and N > 1 this process re-executes itself under torch.distributed.run with N ranks")
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=1, help="BASELINE.json configs[i] (see module docstring)")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the configuration's)")
    ap.add_argument("--prewarm-s", type=float, default=2.0, help="minimum seconds of untimed sampling before the warm-up steps; continues "
                    "(up to 4x as long) until two consecutive 0.5 s windows agree within 1 %% in shader clock and board power")
    ap.add_argument("--no-exact-baseline", action="store_true", help="skip the fp32-bf16x3 (exact 24-bit operand split) timing")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of BASELINE configs[2] and configs[4] that ride on the default line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-torch-baseline", action="store_true")
    ap.add_argument("--no-compile-baseline", action="store_true", help="skip the torch.compile + fp16 autocast form of the PyTorch-ROCm baseline (its compilation takes a minute or two)")
    ap.add_argument("--precision", choices=["fp32", "fp32-bf16x3", "fp16"], default="fp32",
                    help="arithmetic of the convolutions / attention on the matrix pipe (unet.py set_precision).  fp32 (default) and "
                         "fp32-bf16x3 are parity modes: 22-bit fp16 split, 3 products / exact 24-bit bf16 split, 6 products.  fp16 is the "
                         "reduced-precision bulk mode (one fp16 product per MAC, the reference's autocast mode): never the headline")
    ap.add_argument("--seed-base", type=int, default=0, help="first global seed (tests: reproduce one rank's shard alone)")
    ap.add_argument("--dump-samples", default=None, help="directory: every rank saves {seeds, samples} of the timed call (tests)")
    args = ap.parse_args()
    cf = CONFIGS[args.config]
    launch_ranks(args)

    import r2dm_amd
    from r2dm_amd import synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = world > 1
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist:
        import torch.distributed as td

        # RCCL over xGMI ("nccl" IS RCCL on ROCm); R2DM_DIST_BACKEND=gloo lets two ranks share one GPU in tests
        backend = os.environ.get("R2DM_DIST_BACKEND", "nccl")
        td.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    from r2dm_amd.distributed import broadcast_packed_weights, shard_seeds

    B = args.batch or cf["batch"]
    RES = cf["res"]
    ck = synthetic.synthetic_checkpoint(seed=0, resolution=RES)
    ddpm, lidar, _ = r2dm_amd.setup_model(ck, device="cpu", show_info=False, max_batch=B, precision=args.precision)
    ddpm.to(dev)
    broadcast_packed_weights(ddpm.model, dev, src=0)  # rank 0 packs, everyone else adopts the blob
    seeds = shard_seeds(list(range(args.seed_base, args.seed_base + B * world)), rank, world)
    from r2dm_amd.distributed import collective_report
    rccl = collective_report(ddpm.model, dev)  # N > 1: backend, world size, and one all-reduce proving every rank holds the same blob

    def run(steps):
        return ddpm.sample(batch_size=B, num_steps=steps, progress=False, mode=cf["mode"], rng=r2dm_amd.setup_rng(seeds, dev))

    def barrier():
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    # Clock pre-warm (untimed, before the W warm-up steps): an idle MI355X sits at 150 MHz and takes a few hundred
    # milliseconds of load to reach the clock it then sustains; a short default run (16 steps = 0.13 s) measured from cold
    # read 12 % low against the 256-step run of the same job (profiles/r02c).  This only changes the GPU's power state.
    def prewarm(min_s):
        """Untimed sampling until the board is in the state a long sampling call runs in: at least min_s seconds, then until two
        consecutive 0.5 s windows agree within 1 % in median shader clock and board power (at most 4 x min_s)."""
        wins, t_pw = [], time.perf_counter()
        while True:
            with BoardSampler(local) as bs:
                t_w = time.perf_counter()
                while time.perf_counter() - t_w < 0.5:
                    run(8)
                    torch.cuda.synchronize()
            s = bs.summary(all_samples=True)
            wins.append(s)
            el = time.perf_counter() - t_pw
            ok = len(wins) >= 2 and all(wins[-1] and wins[-2] and wins[-1][k] and wins[-2][k] and abs(wins[-1][k] / wins[-2][k] - 1) < 0.01 for k in ("sclk_mhz", "board_w"))
            if (el >= min_s and (ok or wins[-1] is None)) or el >= 4 * min_s:
                return {"seconds": el, "windows": len(wins), "stationary": bool(ok), "last_window": wins[-1]}

    pw = prewarm(args.prewarm_s)
    run(max(args.warmup, 1))  # warm-up: W untimed steps (also sizes the workspace)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # (the board sampler is set up before and torn down after the timed region: its sysfs scan and the join of its 20 ms
    # polling thread used to sit inside it -- ~11 ms per call, 10 % of a 16-step run)
    with BoardSampler(local) as board:
        barrier()
        t0 = time.perf_counter()
        ev0.record()
        out = run(args.steps)
        ev1.record()
        barrier()
        dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if td.get_backend() == "nccl" else "cpu")
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = t.item()
    assert torch.isfinite(out).all()
    gpu_ms = ev0.elapsed_time(ev1)
    if args.dump_samples:
        os.makedirs(args.dump_samples, exist_ok=True)
        torch.save({"seeds": seeds, "samples": out.cpu()}, os.path.join(args.dump_samples, f"rank{rank}.pt"))

    # The convolution launches (~96 % of the FLOPs and of the step time), per kernel class: an extra, untimed pass of a few
    # steps with every conv launch bracketed by HIP events on the sampling stream (r2dm_profile_*).
    conv = None
    if rank == 0:
        ddpm.model.profile_convs(True)
        psteps = min(args.steps, 4)
        run(psteps)
        classes = ddpm.model.read_conv_profile_classes()
        ddpm.model.profile_convs(False)
        ev_pair = ddpm.model.event_pair_overhead()  # (us: around nothing, around an empty kernel)
        NPROD = {"conv_f16x2_kernel": 1 if args.precision == "fp16" else 3, "conv_bf16x3_*": 6}
        WHAT = {"conv_f16x2_kernel": "3x3 implicit-GEMM conv on the fp16 matrix pipe: " +
                                     ("fp16 operands (the fp16 piece of the split alone), 1 product per MAC, fp32 accumulation" if args.precision == "fp16" else
                                      "fp32 operands split to 22 bits (fp16 + 2^11-scaled fp16 residual), 3 products per fp32 product (the "
                                      "residual x residual term, 2^-22 relative, is dropped), two fp32 accumulators per 64-channel output tile, one per "
                                      "128-channel tile (the Cin <= 256 layers with >= 256 such tiles)") +
                                     "; fused GN+SiLU prologue, residual / GroupNorm-statistics epilogue; warp-specialised, persistent",
                "conv_bf16x3_*": "same contract on the bf16 matrix pipe: three bf16 pieces, 6 products per fp32 product",
                "1x1 / in / out convolutions": "proj_f16x2_kernel (1x1 skips and attention projections: split fp16 operands; fp32-input MFMA with precision fp32-bf16x3), conv_few_in_kernel (in_conv), conv_direct_rows_kernel (out_conv)"}
        conv = []
        for name, ms, fl, n in classes:
            if n == 0:
                continue
            e = {"kernel": name, "what": WHAT[name], "launches_per_step": n // psteps, "avg_launch_us": ms * 1e3 / n,
                 "algorithmic_gflop_per_launch": fl / n / 1e9, "tflops": fl / ms / 1e9, "ms_per_step": ms / psteps,
                 "share_of_conv_flops": fl / sum(c[2] for c in classes)}
            if name in NPROD:
                e["products_per_fp32_product"] = NPROD[name]
                e["peak_tflops"] = PEAK_BF16 / NPROD[name] / 1e12
                e["frac"] = e["tflops"] / e["peak_tflops"]
                e["matrix_pipe_tflops"] = e["tflops"] * NPROD[name]  # what the MFMA units actually execute
            else:
                e["peak_tflops"] = PEAK_FP32 / 1e12  # (a mixed class: HBM-bound direct kernels + fp16-pipe projections; the fp32 peak is a yardstick only)
                e["frac"] = e["tflops"] / e["peak_tflops"]
            conv.append(e)
        conv.sort(key=lambda e: -e["ms_per_step"])

    if rank == 0:
        S = cf["sampler_steps"]
        sec_per_step = dt / args.steps
        value = world * B / (sec_per_step * S)
        step_flops = B * cf["flop"] / (gpu_ms / 1e3 / args.steps)
        line = {
            "metric": cf["metric"], "value": value, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": sec_per_step * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16 operands, f32 accumulation / tensors (reduced-precision bulk mode: NOT the parity path)" if args.precision == "fp16" else "f32",
            "dtype_note": {"fp32": "22-bit split fp16 operands, 3 MFMA products, f32 accum",
                           "fp32-bf16x3": "24-bit split bf16 operands, 6 MFMA products, f32 accum",
                           "fp16": "fp16 operands, 1 MFMA product, f32 accum (reduced)"}[args.precision],
            "arithmetic": {"fp32": "fp32 tensors and accumulation; matrix products on 22-bit split fp16 operands (3 MFMA products per fp32 product)",
                           "fp32-bf16x3": "fp32 tensors and accumulation; matrix products on exact 24-bit split bf16 operands (6 MFMA products per fp32 product)",
                           "fp16": "fp32 tensors and accumulation; matrix products on fp16 operands (1 product): reduced precision"}[args.precision],
            "data": "synthetic",
            "config": {"workload": cf["workload"] + f"; timed = one sample() call of --steps reverse steps, value scaled to {S} steps",
                       "baseline_config": args.config, "batch_per_gpu": B, "global_batch": B * world, "resolution": list(RES),
                       "sampler": cf["mode"], "sampler_steps": S, "precision": args.precision, "clock_prewarm": pw,
                       "arithmetic": {"fp32": "22-bit split fp16 operands, 3 products, f32 accum. ", "fp32-bf16x3": "24-bit split bf16 operands, 6 products, f32 accum. ",
                                      "fp16": "fp16 operands, 1 product, f32 accum (reduced). "}[args.precision] +
                                     "fp32 tensors and fp32 accumulation everywhere; the convolutions and the attention core multiply on "
                                     "the 16-bit matrix pipe: " +
                                     {"fp32": "22-bit split operands (fp16 piece + 2^11-scaled fp16 residual; weights pre-scaled per layer by a power "
                                              "of two), 3 products per fp32 product, the residual x residual term (2^-22 relative) dropped, two fp32 "
                                              "accumulators -- fp32-class BY MEASUREMENT (error vs fp64 at or below an fp32 FMA chain's: "
                                              "tests/test_hip_kernels.py, tests/test_hip_range.py), not bit-exact fp32 operands; `exact_split_baseline` "
                                              "is the same job with exact 24-bit operands",
                                      "fp32-bf16x3": "exact 24-bit split operands (three bf16 pieces, 6 products) for the 3x3 convolutions, fp32-input "
                                                     "MFMA for the 1x1 convolutions and the attention core",
                                      "fp16": "fp16 operands, ONE product per MAC (11-bit operands): the reduced-precision bulk mode mirroring the "
                                              "reference's fp16 autocast (sample_and_save.py:70); own tolerance class (tests/test_hip_fp16_mode.py)"}[args.precision],
                       "parallelism": f"dp{world} (independent seeds, no step-loop collective)"},
        }
        dom = conv[0]
        bsum = board.summary()
        line["roofline"] = {
            "bound": "mfma", "achieved": dom["tflops"], "peak": dom["peak_tflops"], "unit": "TFLOP/s", "frac": dom["frac"],
            "traffic": pmc_traffic() if args.config == 1 else None,
            "traffic_measured": False,  # (replayed from the committed PMC passes, see traffic_source: not a measurement of THIS run)
            "traffic_source": "replayed: profiles/conv_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, evidence set %s; "
                              "PMC counters cannot be collected inside the timed process)" % pmc_traffic("source") if args.config == 1 else None,
            "peak_definition": "dominant kernel %s: dense 16-bit MFMA peak 2500 TF/s (2.4 GHz) / %d matrix products per algorithmic "
                               "fp32 product" % (dom["kernel"], dom.get("products_per_fp32_product", 1)),
            "frac_of_round1_peak": dom["tflops"] / (PEAK_BF16 / 6 / 1e12),  # round 1 priced the same algorithmic FLOPs at 2500/6 = 416.7 TF/s
            "frac_of_fp32_mfma_peak": dom["tflops"] * 1e12 / PEAK_FP32,
            "dominant_kernel": dom, "other_conv_kernels": conv[1:],
            "all_convs": {"tflops": sum(e["tflops"] * e["ms_per_step"] for e in conv) / sum(e["ms_per_step"] for e in conv),
                          "ms_per_step": sum(e["ms_per_step"] for e in conv), "launches_per_step": sum(e["launches_per_step"] for e in conv)},
            "whole_step": {"achieved": step_flops / 1e12, "frac_of_fp32_mfma_peak": step_flops / PEAK_FP32,
                           "note": "%.2f GFLOP/image-step x batch / HIP-event time of the timed sample() call" % (cf["flop"] / 1e9)},
            "board": bsum,
            "note": "achieved = algorithmic conv FLOPs / kernel time of the dominant kernel class (HIP events on the sampling "
                    "stream, rank 0, extra untimed pass); peak is the NOMINAL matrix-pipe peak at 2.4 GHz -- the board power "
                    "limit holds the shader clock below that while these kernels run (`board`, sampled during the timed "
                    "region; profiles/r02_power_clock.txt), as it does for hipBLASLt's bf16 GEMM (`hipblaslt_bf16_gemm`); "
                    "traffic = HBM bytes per conv launch from the last committed rocprofv3 PMC passes "
                    "(profiles/conv_traffic.json), config 1 only"}
        # The event pair that brackets a launch spans more than the kernel: reconciled here with rocprofv3's kernel durations (profiles/*_rocprof_summary.txt).
        # `frac` above stays the raw figure (the method of every round); the band is what it becomes with the pair's own span taken off.
        lo, hi = dom["avg_launch_us"] - ev_pair[1], dom["avg_launch_us"] - ev_pair[0]
        line["roofline"]["event_bracket"] = {
            "empty_pair_us": ev_pair[0], "empty_kernel_pair_us": ev_pair[1], "avg_launch_us_raw": dom["avg_launch_us"], "avg_launch_us_band": [lo, hi],
            "frac_band": [dom["frac"] * dom["avg_launch_us"] / hi, dom["frac"] * dom["avg_launch_us"] / lo],
            "note": "an event pair on an idle stream spans empty_pair_us around nothing and empty_kernel_pair_us around an empty kernel (marker processing + one dispatch + "
                    "that kernel's ~1 us), medians of 33 in this process; rocprofv3's average duration of the same kernel lies at the band's lower end (profiles/r06d_*: 93.0 us where this figure was 99.1, band 93.0-94.6; "
                    "r06b: 94.8 against a band of 96.1-97.7 -- a pair costs up to ~1.5 us more on a busy stream than on an idle one)"}
        if bsum and dom.get("products_per_fp32_product"):
            pk = PEAK_BF16 * bsum["sclk_mhz"] / 2400.0 / dom["products_per_fp32_product"] / 1e12
            line["roofline"]["peak_at_sustained_clock"] = pk
            line["roofline"]["frac_at_sustained_clock"] = dom["tflops"] / pk
        if world == 1 and not args.no_torch_baseline:
            line["roofline"]["hipblaslt_bf16_gemm"] = hipblaslt_reference(dev)
        if world == 1 and not args.no_torch_baseline:
            tb = torch_rocm_baseline(ck, cf, B, dev, compiled=not args.no_compile_baseline)
            tb["speedup"] = value / tb["value"]
            line["torch_rocm_baseline"] = tb
            # north_star: ">= N x the reference single-GPU PyTorch sampler" -- BASELINE.md / BASELINE.json ("published": {}) hold no
            # published number for this metric, so `vs_baseline` stays null (the bench contract); the ratio against the baseline
            # measured beside it in this run is reported under its own name
            line["vs_torch_rocm_baseline"] = value / tb["value"]
            line["vs_torch_rocm_baseline_definition"] = "value / torch_rocm_baseline.value (the reference's sampler as stock PyTorch-ROCm runs it on this GPU, fp32, same run)"
            ac = tb.get("fp16_autocast", {})
            if "value" in ac:
                ac["speedup"] = value / ac["value"]
                if args.precision == "fp16":  # the reduced mode is compared with the reference's reduced mode
                    line["vs_torch_rocm_baseline"] = value / ac["value"]
                    line["vs_torch_rocm_baseline_definition"] = "value / torch_rocm_baseline.fp16_autocast.value (the reference's fp16-autocast bulk mode on this GPU, same run)"
            cc = tb.get("compiled_fp16_autocast", {})
            if "value" in cc:
                cc["speedup"] = value / cc["value"]
                if args.precision == "fp16" and cc["value"] > ac.get("value", 0.0):  # (against the FASTER form of the reference's bulk mode)
                    line["vs_torch_rocm_baseline"] = value / cc["value"]
                    line["vs_torch_rocm_baseline_definition"] = "value / torch_rocm_baseline.compiled_fp16_autocast.value (torch.compile + fp16 autocast: the reference's bulk sampler as sample_and_save.py runs it, same run)"
        if world == 1 and not args.no_exact_baseline and args.precision == "fp32":
            ddpm.model.set_precision("fp32-bf16x3")
            prewarm(1.0)
            run(max(args.warmup, 1))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run(args.steps)
            torch.cuda.synchronize()
            dte = time.perf_counter() - t1
            ddpm.model.set_precision("fp32")
            line["exact_split_baseline"] = {"value": B / (dte / args.steps * S), "unit": "images/s", "ms_per_step": dte / args.steps * 1e3,
                                            "precision": "fp32-bf16x3", "headline_speedup": value / (B / (dte / args.steps * S)),
                                            "what": "the same timed sample() call with exact 24-bit operands: three bf16 pieces, six products per fp32 "
                                                    "product (conv_bf16x3.hip), fp32-input MFMA for 1x1 convolutions and attention"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(ck, cf)
        if world == 1 and args.config == 1 and args.precision == "fp32" and args.batch is None and not args.no_other_configs:
            # BASELINE configs[2] and configs[4] (per-GPU shard) beside the headline, same process and board state: one timed
            # sample() call each after a 1 s pre-warm (their full lines incl. baselines: `--config 2` / `--config 4`)
            del ddpm
            torch.cuda.empty_cache()
            others = {}
            for ci, osteps in ((2, 32), (4, 8)):
                oc = CONFIGS[ci]
                ock = synthetic.synthetic_checkpoint(seed=0, resolution=oc["res"])
                om, _, _ = r2dm_amd.setup_model(ock, device=dev, show_info=False, max_batch=oc["batch"], precision=args.precision)
                oseeds = list(range(oc["batch"]))
                orun = lambda n: om.sample(batch_size=oc["batch"], num_steps=n, progress=False, mode=oc["mode"], rng=r2dm_amd.setup_rng(oseeds, dev))
                t_pw = time.perf_counter()
                while time.perf_counter() - t_pw < 1.0:
                    orun(4)
                    torch.cuda.synchronize()
                orun(osteps)  # (same step count as the timed call: its schedule table is then cached, as for any repeated call)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                oo = orun(osteps)
                torch.cuda.synchronize()
                odt = time.perf_counter() - t1
                assert torch.isfinite(oo).all()
                others["configs[%d]" % ci] = {"workload": oc["workload"], "metric": oc["metric"], "value": oc["batch"] / (odt / osteps * oc["sampler_steps"]),
                                              "unit": "images/s per GPU", "ms_per_step": odt / osteps * 1e3, "steps": osteps}
                del om
                torch.cuda.empty_cache()
            line["other_configs"] = others
        line["rccl"] = rccl
        print(json.dumps(line))
    if dist:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
