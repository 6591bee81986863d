"""r2dm_amd -- MI355X-native R2DM sampler (DDPM/DDIM reverse process over the Efficient U-Net).

Drop-in for the sampling API of kazuto1011/r2dm; the per-step hot path runs as hand-written HIP
kernels (``libr2dm_hip.so``, C ABI in ``include/r2dm_hip.h``).  There is no CPU fallback.
"""
from .diffusion import ContinuousTimeGaussianDiffusion, DiscreteTimeGaussianDiffusion, GaussianDiffusion
from .inference import setup_model, setup_rng
from .lidar import LiDARUtility
from .option import Config
from .unet import EfficientUNet

__all__ = [
    "ContinuousTimeGaussianDiffusion", "DiscreteTimeGaussianDiffusion", "GaussianDiffusion", "EfficientUNet",
    "LiDARUtility", "Config", "setup_model", "setup_rng",
]
__version__ = "0.4.0"
