"""Configuration schema stored inside checkpoints (``ckpt["cfg"]``).

Field-for-field compatible with /root/reference/utils/option.py:6-77 so that
``Config(**ckpt["cfg"])`` accepts the dicts the reference's training script writes
(/root/reference/train.py:294-303).  Plain dataclasses: nested dicts are coerced by hand, no
pydantic dependency.  ``min_depth`` / ``max_depth`` are class attributes, not fields, exactly as
upstream (they are therefore absent from saved configs).
"""
from __future__ import annotations

from dataclasses import dataclass, field, fields, is_dataclass
from typing import Optional, Tuple


def _coerce(cls, value):
    if isinstance(value, cls):
        return value
    if value is None:
        return cls()
    if isinstance(value, dict):
        known = {f.name for f in fields(cls)}
        unknown = set(value) - known
        if unknown:
            raise TypeError(f"{cls.__name__}: unexpected fields {sorted(unknown)}")
        return cls(**value)
    raise TypeError(f"cannot build {cls.__name__} from {type(value).__name__}")


@dataclass
class ModelConfig:
    architecture: str = "efficient_unet"
    base_channels: int = 64
    temb_channels: Optional[int] = None
    channel_multiplier: Tuple[int, int, int, int] = (1, 2, 4, 8)
    num_residual_blocks: Tuple[int, int, int, int] = (3, 3, 3, 3)
    gn_num_groups: int = 32 // 4
    gn_eps: float = 1e-6
    attn_num_heads: int = 8
    coords_encoding: Optional[str] = "fourier_features"
    dropout: float = 0.0

    def __post_init__(self):
        self.channel_multiplier = tuple(self.channel_multiplier)
        self.num_residual_blocks = tuple(self.num_residual_blocks)


@dataclass
class DiffusionConfig:
    num_training_steps: Optional[int] = None
    num_sampling_steps: int = 1024
    prediction_type: str = "eps"
    loss_type: str = "l2"
    noise_schedule: str = "cosine"
    timestep_type: str = "continuous"

    def __post_init__(self):
        if self.prediction_type not in ("eps", "v", "x_0"):
            raise ValueError(f"prediction_type: {self.prediction_type!r}")
        if self.timestep_type not in ("continuous", "discrete"):
            raise ValueError(f"timestep_type: {self.timestep_type!r}")


@dataclass
class TrainingConfig:
    batch_size_train: int = 8
    batch_size_eval: int = 8
    num_workers: int = 4
    num_steps: int = 300_000
    steps_save_image: int = 5_000
    steps_save_model: int = 10_000
    gradient_accumulation_steps: int = 1
    lr: float = 1e-4
    lr_warmup_steps: int = 10_000
    adam_beta1: float = 0.9
    adam_beta2: float = 0.99
    adam_weight_decay: float = 0.0
    adam_epsilon: float = 1e-8
    ema_decay: float = 0.995
    ema_update_every: int = 10
    mixed_precision: Optional[str] = "fp16"
    dynamo_backend: Optional[str] = "inductor"
    output_dir: str = "logs/diffusion"
    seed: int = 0


@dataclass
class DataConfig:
    dataset: str = "kitti_360"
    depth_format: str = "log_depth"
    projection: str = "spherical-1024"
    train_depth: bool = True
    train_reflectance: bool = True
    resolution: Tuple[int, int] = (64, 1024)
    min_depth = 1.45
    max_depth = 80.0

    def __post_init__(self):
        self.resolution = tuple(self.resolution)
        if self.depth_format not in ("log_depth", "inverse_depth", "depth"):
            raise ValueError(f"depth_format: {self.depth_format!r}")


@dataclass
class Config:
    data: DataConfig = field(default_factory=DataConfig)
    model: ModelConfig = field(default_factory=ModelConfig)
    diffusion: DiffusionConfig = field(default_factory=DiffusionConfig)
    training: TrainingConfig = field(default_factory=TrainingConfig)

    def __post_init__(self):
        self.data = _coerce(DataConfig, self.data)
        self.model = _coerce(ModelConfig, self.model)
        self.diffusion = _coerce(DiffusionConfig, self.diffusion)
        self.training = _coerce(TrainingConfig, self.training)
