#!/usr/bin/env python3
"""Drop-in for the reference's bulk sampler (/root/reference/sample_and_save.py): same CLI, same per-seed output
files ``samples_{seed:010d}.pth`` holding a (5,H,W) [depth, x, y, z, reflectance] tensor.

Multi-GPU: launch with ``python -m torch.distributed.run --nproc-per-node N sample_and_save.py ...`` -- one process
per MI355X, seeds sharded contiguously (what Accelerate's split_batches DataLoader does in the reference,
:25-46), weights packed once on rank 0 and broadcast over RCCL/xGMI, no collective in the sampling loop.
Precision: fp32 parity by default (``--precision fp32``: 22-bit split fp16 operands, three MFMA products per fp32 product;
``fp32-bf16x3``: three bf16 pieces, fp32 operand range).  The reference runs this script under fp16 autocast (:70,
utils/option.py:49); ``--precision fp16`` is that bulk mode here: one fp16 product per MAC, fp32 accumulation and tensors,
its own tolerance class (tests/test_hip_fp16_mode.py).
``--max-batch`` / the batch size fix the layer tilings: per-seed results are bit-reproducible for a fixed batch size only."""
import os
from argparse import ArgumentParser
from pathlib import Path

import torch

import r2dm_amd
from r2dm_amd.distributed import broadcast_packed_weights, shard_seeds


def sample(args):
    torch.set_grad_enabled(False)
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", 1), ("RANK", 0), ("LOCAL_RANK", 0)))
    local = local % max(torch.cuda.device_count(), 1)  # (tests: several ranks share one GPU)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as td

        # RCCL over xGMI ("nccl" IS RCCL on ROCm); R2DM_DIST_BACKEND=gloo lets two ranks share one GPU in tests (as bench.py)
        backend = os.environ.get("R2DM_DIST_BACKEND", "nccl")
        td.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))

    ddpm, lidar_utils, cfg = r2dm_amd.setup_model(args.ckpt, show_info=rank == 0, max_batch=args.max_batch or args.batch_size,
                                                  precision=args.precision)
    ddpm.to(device)
    lidar_utils.to(device)
    broadcast_packed_weights(ddpm.model, device, src=0)

    save_dir = Path(args.output_dir)
    if rank == 0:
        save_dir.mkdir(parents=True, exist_ok=True)
    if world > 1:
        td.barrier()

    mine = shard_seeds(list(range(args.num_samples)), rank, world)
    for i in range(0, len(mine), args.batch_size):
        seeds = mine[i: i + args.batch_size]
        samples = ddpm.sample(batch_size=len(seeds), num_steps=args.num_steps, mode=args.mode,
                              rng=r2dm_amd.setup_rng(seeds, device=device), progress=False).clamp(-1, 1)
        samples = lidar_utils.postprocess(samples)  # denormalize -> revert_depth -> to_xyz -> concat, one kernel
        for seed, s in zip(seeds, samples):
            torch.save(s.clone(), save_dir / f"samples_{seed:010d}.pth")
    if world > 1:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    parser = ArgumentParser()
    parser.add_argument("--ckpt", type=str)
    parser.add_argument("--output_dir", type=str)
    parser.add_argument("--batch_size", type=int, default=64)
    parser.add_argument("--num_samples", type=int, default=10_000)
    parser.add_argument("--num_steps", type=int, default=256)
    parser.add_argument("--mode", choices=["ddpm", "ddim"], default="ddpm")
    parser.add_argument("--precision", choices=["fp32", "fp32-bf16x3", "fp16"], default="fp32")
    parser.add_argument("--max-batch", type=int, default=0, help="extension: the batch size the layer tilings are planned for (default: --batch_size)")
    sample(parser.parse_args())
