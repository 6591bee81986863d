#!/usr/bin/env python3
"""Drop-in for the tensor-producing half of the reference's generate.py (/root/reference/generate.py:16-42):
same CLI, same sampling call, same post-processing of the returned (S+1,B,2,H,W) stack.  The PNG / BEV / mp4
rendering half (generate.py:44-76; kornia, imageio, torchvision) is out of scope: tensors are saved instead."""
import argparse
from pathlib import Path

import torch

import r2dm_amd


def main(args):
    torch.set_grad_enabled(False)
    ddpm, lidar_utils, _ = r2dm_amd.setup_model(args.ckpt, device=args.device, max_batch=args.batch_size)
    xs = ddpm.sample(batch_size=args.batch_size, num_steps=args.sampling_steps, mode=args.mode, return_all=True).clamp(-1, 1)
    xs = lidar_utils.denormalize(xs)
    xs[:, :, [0]] = lidar_utils.revert_depth(xs[:, :, [0]]) / lidar_utils.max_depth
    points = lidar_utils.postprocess(_last_sample_normalized(xs, lidar_utils))
    torch.save({"frames": xs.cpu(), "points": points.cpu()}, args.output)
    print(f"saved {tuple(xs.shape)} frames and {tuple(points.shape)} [depth,x,y,z,reflectance] maps to {args.output}")


def _last_sample_normalized(xs, lidar_utils):
    """Undo the display scaling of the final frame to feed the fused xyz post-processing kernel."""
    last = xs[-1].clone()
    last[:, [0]] = lidar_utils.convert_depth(last[:, [0]] * lidar_utils.max_depth)
    return lidar_utils.normalize(last)


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--ckpt", type=Path, required=True)
    parser.add_argument("--device", choices=["cuda"], default="cuda")
    parser.add_argument("--mode", choices=["ddpm", "ddim"], default="ddpm")
    parser.add_argument("--batch_size", type=int, default=1)
    parser.add_argument("--sampling_steps", type=int, default=256)
    parser.add_argument("--output", type=Path, default=Path("samples.pt"))
    args = parser.parse_args()
    args.device = torch.device(args.device)
    main(args)
