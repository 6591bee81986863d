"""torch.hub entry point with the reference's signature (/root/reference/hubconf.py:21-37):

    ddpm, lidar_utils, cfg = torch.hub.load("<this repo>", "pretrained_r2dm", device="cuda")
    x = ddpm.sample(batch_size=8, num_steps=256)          # MI355X HIP kernels, see r2dm_amd/

Only the R2DM sampler is provided; the reference's bonus RangeNet++ entry points (hubconf.py:45-104) belong to
the evaluation pipeline and are out of scope."""
dependencies = ["torch", "numpy"]


def _get_r2dm_url(key: str) -> str:
    return f"https://github.com/kazuto1011/r2dm/releases/download/weights/{key}.pth"


def pretrained_r2dm(config: str = "r2dm-h-kitti360-300k", ckpt: str = None, **kwargs):
    """R2DM sampler from a released checkpoint.

    Args:
        config: release key of the pre-trained weights (default "r2dm-h-kitti360-300k").
        ckpt:   path to (or dict of) a checkpoint; if given, `config` is ignored.
        **kwargs: forwarded to ``r2dm_amd.setup_model`` (device=..., ema=..., show_info=..., max_batch=...).
    Returns:
        (ddpm, lidar_utils, cfg) exactly like the reference.
    """
    from r2dm_amd import setup_model

    if ckpt is None:
        from torch.hub import load_state_dict_from_url

        ckpt = load_state_dict_from_url(_get_r2dm_url(config), map_location="cpu")
    return setup_model(ckpt, **kwargs)
