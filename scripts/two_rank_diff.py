#!/usr/bin/env python3
"""GPU-box diagnostic (round 5): the two-rank drop-in scenario of tests/test_dropin_scripts.py::test_sample_and_save_two_ranks_on_one_gpu, REPS times,
reporting per repetition which of the five sample files differ between the two-rank and the single-process run of sample_and_save.py and by how much
(library: R2DM_HIP_LIB or the default)."""
import os, socket, subprocess, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synthetic_ckpt
reps = int(os.environ.get("REPS", "6"))
tmp = tempfile.mkdtemp()
ck = os.path.join(tmp, "synthetic.pth"); torch.save(synthetic_ckpt(), ck)
args = ["--ckpt", ck, "--batch_size", "2", "--num_samples", "5", "--num_steps", "2"]
one = os.path.join(tmp, "one")
subprocess.run([sys.executable, "sample_and_save.py", "--output_dir", one] + args, cwd=ROOT, check=True, capture_output=True, timeout=600)
subprocess.run([sys.executable, "sample_and_save.py", "--output_dir", one + "b"] + args, cwd=ROOT, check=True, capture_output=True, timeout=600)
names = [f"samples_{i:010d}.pth" for i in range(5)]
print("single process twice: files differing", sum(int(not torch.equal(torch.load(os.path.join(one, n)), torch.load(os.path.join(one + 'b', n)))) for n in names), flush=True)
bad = 0
for r in range(reps):
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    two = os.path.join(tmp, f"two{r}")
    env = dict(os.environ, R2DM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        "sample_and_save.py", "--output_dir", two] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    if p.returncode:
        print("rep", r, "two-rank run failed:", p.stderr[-300:]); bad += 1; continue
    d = []
    for n in names:
        a, b = torch.load(os.path.join(two, n)).float().cpu(), torch.load(os.path.join(one, n)).float().cpu()
        if not torch.equal(a, b):
            df = (a - b).abs()
            d.append(f"{n[-6:-4]}: max {df.max().item():.3g} px {int((df > 0).sum())} of {df.numel()} (channels differing {[int((df[c] > 0).sum()) for c in range(a.shape[0])]})")
    bad += int(bool(d))
    print("rep", r, "differing:", d if d else "none", flush=True)
print(f"two_rank_diff: {bad} of {reps} repetitions differ (library {os.environ.get('R2DM_HIP_LIB', 'default')})")
