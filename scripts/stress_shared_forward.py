#!/usr/bin/env python3
"""GPU-box probe: is the U-Net forward independent of a second PROCESS running the same forward on the same GPU?

ROLE=hog runs forwards until it is killed (it touches READY_FILE once its first forwards have completed); the main role
repeats one forward ITERS times alone and ITERS times next to the hog and counts outputs that differ bitwise from the
first one.  It also reports the time per forward of both phases: the slowdown is the evidence that the two processes
really shared the GPU during the second phase.  HOG_ENV="K=V ..." adds variables to the hog's environment."""
import os, subprocess, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r2dm_amd
from r2dm_amd import synthetic
dev = torch.device("cuda", 0)
B = int(os.environ.get("B", "2"))
ck = synthetic.synthetic_checkpoint(seed=0, resolution=(64, 1024))
ddpm, _, _ = r2dm_amd.setup_model(ck, device=dev, show_info=False, max_batch=B, precision=os.environ.get("PRECISION", "fp32"))
g = torch.Generator(device=dev).manual_seed(5)
x = torch.randn(B, 2, 64, 1024, device=dev, generator=g); c = torch.full((B,), -3.0, device=dev)
if os.environ.get("ROLE") == "hog":
    t0 = time.time()
    with ddpm.model.deferred_range_check():
        while time.time() - t0 < float(os.environ.get("SECS", "600")):
            for _ in range(10): ddpm.model(x, c)
            torch.cuda.synchronize()
            if os.environ.get("READY_FILE"): open(os.environ["READY_FILE"], "w").close()
    sys.exit(0)
iters = int(os.environ.get("ITERS", "150"))
for hog in ("none", "forward"):
    p = None
    if hog != "none":
        ready = tempfile.mktemp(prefix="hog_ready_")
        henv = dict(os.environ, ROLE="hog", READY_FILE=ready)
        for kv in os.environ.get("HOG_ENV", "").split():  # e.g. HOG_ENV="R2DM_CONV_ALGO=f32 HSA_CU_MASK=0:128-255"
            k, v = kv.split("=", 1); henv[k] = v
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=henv)
        t0 = time.time()
        while not os.path.exists(ready) and time.time() - t0 < 240 and p.poll() is None: time.sleep(0.5)
        assert os.path.exists(ready), "the neighbour process did not come up"
    ref = ddpm.model(x, c).clone()
    bad = 0; worst = 0.0
    torch.cuda.synchronize(); t0 = time.time()
    with ddpm.model.deferred_range_check():
        for i in range(iters):
            y = ddpm.model(x, c)
            if not torch.equal(y, ref):
                bad += 1; worst = max(worst, (y - ref).abs().max().item())
    ms = (time.time() - t0) / iters * 1e3
    alive = p is None or p.poll() is None
    print(f"[{os.environ.get('TAG','')}] B={B} precision={os.environ.get('PRECISION','fp32')} algo={os.environ.get('R2DM_CONV_ALGO','default')} hog_env={os.environ.get('HOG_ENV','')!r} "
          f"neighbour={hog:8s}: {bad:3d} of {iters} forwards differ from the first (max |diff| {worst:.2e}); {ms:.2f} ms per forward+compare"
          f"{'' if alive else '  [NEIGHBOUR DIED EARLY]'}", flush=True)
    if p is not None:
        p.terminate(); p.wait()
