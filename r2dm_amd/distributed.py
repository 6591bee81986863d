"""Data-parallel sampling over independent seeds.

The reference shards a DataLoader of *seeds* across ranks with Accelerate
(/root/reference/sample_and_save.py:25-46) and every rank re-reads the checkpoint from disk.  Here:
one process per GPU, contiguous seed shards, ONE RCCL broadcast of the packed weight blob over
xGMI at start-up, and no collective inside the step loop (samples are independent).  Results are
partition-invariant because every sample owns a generator seeded by its global seed
(/root/reference/utils/inference.py:113-114, /root/reference/models/diffusion/base.py:81-85).
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import torch


def _dist():
    import torch.distributed as td

    return td if (td.is_available() and td.is_initialized()) else None


def shard_seeds(seeds: Sequence[int], rank: int, world: int) -> List[int]:
    """Contiguous split; the first (len % world) ranks get one extra seed (even_batches=False)."""
    n = len(seeds)
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return list(seeds[lo: lo + base + (1 if rank < extra else 0)])


def broadcast_tensor(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    td = _dist()
    if td is not None and td.get_world_size() > 1:
        td.broadcast(t, src=src)
    return t


def broadcast_packed_weights(model, device, src: int = 0) -> None:
    """Rank `src` packs its weights into the engine blob; the blob is broadcast and adopted by all.

    Under RCCL ("nccl") the blob travels GPU to GPU over xGMI.  Under gloo (tests: several ranks on one GPU) it is staged
    through host memory -- gloo's device support is not a given on a ROCm build."""
    td = _dist()
    if td is None or td.get_world_size() == 1:
        model.packed_weights(device)
        return
    on_host = td.get_backend() != "nccl"
    meta = [model.packed_layout_hash() if td.get_rank() == src else None]
    # (the layout the blob was packed for: checked by every adopter.  Under RCCL the pickled object travels through a device buffer: name the
    # device -- the default is the CURRENT device, which is cuda:0 on every rank of a caller that did not torch.cuda.set_device(local_rank))
    td.broadcast_object_list(meta, src=src, **({} if on_host else {"device": torch.device(device)}))
    if td.get_rank() == src:
        blob = model.packed_weights(device)
        wire = blob.cpu() if on_host else blob
    else:
        wire = torch.empty(model.packed_weight_bytes(), dtype=torch.uint8, device="cpu" if on_host else device)
    td.broadcast(wire, src=src)
    if td.get_rank() != src:
        model.adopt_packed_weights(wire.to(device) if on_host else wire, layout_hash=meta[0])


def blob_checksum(blob: torch.Tensor) -> int:
    """A 63-bit checksum of a packed weight blob, computed where the blob lives: the wrapping sum of its 8-byte words, each multiplied by an
    odd weight that depends on its position (word i times 2 i + 1) -- a plain sum would not notice two words changing places."""
    blob = blob.reshape(-1)
    if blob.dtype != torch.uint8:
        blob = blob.view(torch.uint8)
    if blob.storage_offset() % 8 or not blob.is_contiguous():  # (a view into a larger buffer: int64 views want 8-byte alignment)
        blob = blob.contiguous().clone()
    n = blob.numel() // 8 * 8
    words = blob[:n].view(torch.int64)
    weight = torch.arange(words.numel(), dtype=torch.int64, device=words.device) * 2 + 1
    tail = int((blob[n:].to(torch.int64) * torch.arange(1, blob.numel() - n + 1, dtype=torch.int64, device=blob.device)).sum().item()) if n < blob.numel() else 0
    return (int((words * weight).sum().item()) + tail) & 0x7FFFFFFFFFFFFFFF


def collective_report(model=None, device=None) -> dict:
    """What a multi-GPU line must prove: which backend ran, over how many ranks, and that every rank samples
    from the SAME weights -- one all-reduce (MIN and MAX) of the adopted blob's checksum over the job's process group (RCCL over xGMI
    under "nccl").  Single process: world_size 1, no collective."""
    td = _dist()
    if td is None or td.get_world_size() == 1:
        return {"backend": None, "world_size": 1, "blob_crc_equal_on_all_ranks": None}
    crc = blob_checksum(model.packed_weights(device)) if model is not None else 0
    on_host = td.get_backend() != "nccl"
    t = torch.tensor([crc, -crc], dtype=torch.int64, device="cpu" if on_host else device)
    td.all_reduce(t, op=td.ReduceOp.MIN)  # (min(crc), min(-crc) = -max(crc))
    lo, hi = int(t[0].item()), -int(t[1].item())
    return {"backend": td.get_backend() + (" (RCCL)" if td.get_backend() == "nccl" else ""), "world_size": td.get_world_size(),
            "blob_crc_equal_on_all_ranks": lo == hi == crc, "blob_crc": crc}


def sample_sharded(sample_fn: Callable[[List[int]], torch.Tensor], seeds: Sequence[int], gather: bool = True,
                   device=None):
    """Run ``sample_fn(my_seeds) -> (n_local, ...)`` on this rank's shard; optionally all-gather the
    per-seed results (in global seed order) on every rank.  ``device``: where the gather buffers live (default: the
    current CUDA device under the NCCL/RCCL backend, CPU under gloo) -- it must be the same kind on every rank, also
    on ranks whose shard is empty (fewer seeds than ranks)."""
    td = _dist()
    world = td.get_world_size() if td else 1
    rank = td.get_rank() if td else 0
    seeds = list(seeds)
    mine = shard_seeds(seeds, rank, world)
    out = sample_fn(mine) if mine else None
    if not gather or world == 1 or not seeds:
        return out, mine
    sizes = [len(shard_seeds(seeds, r, world)) for r in range(world)]
    if device is None:
        device = (torch.device("cuda", torch.cuda.current_device()) if td.get_backend() == "nccl" else torch.device("cpu"))
    device = torch.device(device)
    # rank 0 always owns a non-empty shard (the first len % world ranks get the extra seeds): it announces shape + dtype
    meta = [(tuple(out.shape[1:]), out.dtype) if rank == 0 else None]
    td.broadcast_object_list(meta, src=0, **({"device": device} if td.get_backend() == "nccl" else {}))
    tail, dtype = meta[0]
    buf = torch.zeros(max(sizes), *tail, device=device, dtype=dtype)
    if out is not None:
        buf[: len(mine)] = out.to(device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    td.all_gather(parts, buf)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)]), mine
