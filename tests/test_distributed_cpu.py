"""CPU, world_size 2, gloo: the data-parallel sharding layer (r2dm_amd/distributed.py).

The denoiser itself cannot run without a GPU (no CPU fallback in the product), so the per-rank sampler here is
the ORACLE on a tiny stand-in problem -- allowed in tests -- which exercises exactly the code bench.py and
sample_and_save.py run around the HIP sampler: contiguous seed shards, one broadcast of a packed blob, per-seed
generators, gather in seed order.  Checked: results are independent of the number of ranks."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tiny_sampler(blob):
    """A 3-step 'diffusion' whose denoiser is a fixed random linear map taken from the broadcast blob."""
    from oracle import r2dm_oracle as O

    w = blob.view(torch.float32)[: 2 * 2 * 9].reshape(2, 2, 3, 3)

    def fn(seeds):
        rng = [torch.Generator().manual_seed(int(s)) for s in seeds]
        net = lambda x, c: O.conv_ring(x, w, None) * 0.1 + c[:, None, None, None] * 0.01
        return O.sample_continuous(net, (len(seeds), 2, 8, 32), 3, rng=rng)

    return fn


def _worker(rank, world, port, seeds, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as td

    from r2dm_amd import distributed as D

    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        blob = torch.zeros(4096, dtype=torch.uint8)
        if rank == 0:  # only the source rank has the "weights"
            blob = torch.randn(1024, generator=torch.Generator().manual_seed(42)).view(torch.uint8).clone()
        D.broadcast_tensor(blob, src=0)
        out, mine = D.sample_sharded(_tiny_sampler(blob), seeds, gather=True)
        assert mine == D.shard_seeds(seeds, rank, world)
        if rank == 0:
            torch.save(out, out_path)
    finally:
        td.destroy_process_group()


def test_shard_seeds_partition():
    from r2dm_amd.distributed import shard_seeds

    for n in (0, 1, 5, 8, 64, 67):
        for world in (1, 2, 3, 8):
            parts = [shard_seeds(list(range(n)), r, world) for r in range(world)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
    assert shard_seeds(list(range(64)), 3, 8) == list(range(24, 32))  # BASELINE configs[3]: 8 seeds per GPU


@pytest.mark.parametrize("seeds", [[0, 1, 2, 3, 4, 5], [10, 11, 12, 13, 14], [7]])  # [7]: rank 1 has an empty shard
def test_two_ranks_match_single_process(tmp_path, seeds):
    blob = torch.randn(1024, generator=torch.Generator().manual_seed(42)).view(torch.uint8).clone()
    want = _tiny_sampler(blob)(seeds)
    out_path = str(tmp_path / "out.pt")
    mp.spawn(_worker, args=(2, _free_port(), seeds, out_path), nprocs=2, join=True)
    got = torch.load(out_path)
    assert got.shape == want.shape
    assert torch.equal(got, want)  # partition-invariant: per-seed generators, no cross-sample coupling


def _worker8(rank, world, port, seeds, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as td

    from r2dm_amd import distributed as D

    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        blob = torch.zeros(4096, dtype=torch.uint8)
        if rank == 0:
            blob = torch.randn(1024, generator=torch.Generator().manual_seed(42)).view(torch.uint8).clone()
        D.broadcast_tensor(blob, src=0)

        class Holder:  # what collective_report needs of a model: the adopted blob
            def packed_weights(self, device):
                return blob

        rep = D.collective_report(Holder(), "cpu")
        assert rep["world_size"] == world and rep["backend"].startswith("gloo") and rep["blob_crc_equal_on_all_ranks"] is True
        bad = D.collective_report(type("H", (), {"packed_weights": lambda self, d: blob + (1 if rank == world - 1 else 0)})(), "cpu")
        assert bad["blob_crc_equal_on_all_ranks"] is False  # one rank with other weights is seen by every rank
        out, mine = D.sample_sharded(_tiny_sampler(blob), seeds, gather=True)
        assert mine == D.shard_seeds(seeds, rank, world)
        if rank == 0:
            torch.save(out, out_path)
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("seeds", [list(range(16)), list(range(100, 113)), [3, 1, 4]])  # even (2 per rank) / uneven (13 over 8) / ranks 3..7 empty
def test_eight_ranks_match_single_process(tmp_path, seeds):
    """the sharding layer at the world size of BASELINE configs[3] / configs[4] (8 ranks, gloo on CPU): contiguous
    seed shards incl. uneven and EMPTY ones, one broadcast, a gather in seed order -- equal to one process sampling all seeds; and the
    checksum all-reduce bench.py's `rccl` field reports (equal blobs: True on every rank; one deviating rank: False on every rank)."""
    blob = torch.randn(1024, generator=torch.Generator().manual_seed(42)).view(torch.uint8).clone()
    # (the stand-in denoiser is a CPU library convolution whose algorithm -- and last bits -- depend on the batch it is called with;
    # what this test pins is the sharding layer: every shard sampled as one process would sample that shard, gathered in seed order)
    from r2dm_amd.distributed import shard_seeds

    want = torch.cat([_tiny_sampler(blob)(sh) for sh in (shard_seeds(seeds, r, 8) for r in range(8)) if sh])
    assert torch.allclose(want, _tiny_sampler(blob)(seeds), atol=1e-5)
    out_path = str(tmp_path / "out.pt")
    mp.spawn(_worker8, args=(8, _free_port(), seeds, out_path), nprocs=8, join=True)
    got = torch.load(out_path)
    assert got.shape == want.shape and torch.equal(got, want)


def test_broadcast_packed_weights_single_process_is_a_noop_pack():
    """Without an initialised process group the helper only packs locally (needs the GPU -> here just the error)."""
    import r2dm_amd
    from r2dm_amd import _lib, distributed as D
    from conftest import GOLDEN_RES, synthetic_ckpt

    ddpm, _, _ = r2dm_amd.setup_model(synthetic_ckpt(resolution=GOLDEN_RES), device="cpu", show_info=False)
    with pytest.raises(_lib.R2DMError):
        D.broadcast_packed_weights(ddpm.model, torch.device("cpu"))
    assert ddpm.model.packed_weight_bytes() > 100e6
