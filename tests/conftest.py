import os
import sys
import warnings

import numpy as np
import pytest
import torch

# the fp32 / fp64 CPU oracles (torch + oneDNN) get SLOWER beyond a few dozen threads on the 256-core GPU hosts
torch.set_num_threads(min(32, os.cpu_count() or 1))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore", category=UserWarning)

GOLDEN = os.path.join(ROOT, "tests", "golden")
GOLDEN_RES = (16, 128)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests are skipped (not failed) where there is no ROCm device or the HIP library has not been built, so a
    plain `pytest` is green on a CPU-only machine.  On a GPU box with the library missing they FAIL loudly instead:
    the product has no CPU fallback and a silent skip there would hide a broken build."""
    lib = os.path.join(ROOT, "r2dm_amd", "libr2dm_hip.so")
    if torch.cuda.is_available():
        if not os.path.exists(lib):
            raise pytest.UsageError(f"{lib} is missing on a GPU box: run `python __graft_entry__.py` (build) first")
        return
    skip = pytest.mark.skip(reason="needs a ROCm GPU (MI355X): run with -m gpu on the GPU box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get


_CKPT = {}


def synthetic_ckpt(**kw):
    """Cached synthetic checkpoints (regenerated deterministically, never stored)."""
    from r2dm_amd import synthetic

    key = tuple(sorted(kw.items()))
    if key not in _CKPT:
        _CKPT[key] = synthetic.synthetic_checkpoint(seed=0, **kw)
    return _CKPT[key]


def rnd(seed, *shape):
    return torch.from_numpy(np.random.Generator(np.random.PCG64(seed)).standard_normal(shape).astype(np.float32))


def max_abs(a, b):
    return (a.double() - b.double()).abs().max().item()
