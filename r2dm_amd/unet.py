"""Efficient U-Net denoiser whose forward pass is the HIP engine.

Drop-in for the reference class (/root/reference/models/efficient_unet.py:188-295): same
constructor arguments, same attributes (``resolution``, ``in_channels``, ``out_channels``,
``coords``), same 268-key state dict, same ``forward(images (B,C,H,W), timesteps (B,))``.

Structure here is deliberately NOT the reference's module zoo: parameters are registered from the
flat table in ``spec.py`` on anonymous containers (so ``state_dict()`` / ``load_state_dict`` /
``.to()`` keep working), and ``forward`` is one call into ``libr2dm_hip.so``
(``r2dm_unet_forward``) which enqueues every kernel of the pass on the current stream.  The
engine keeps its own packed copy of the weights (the "blob"); it is rebuilt lazily whenever the
parameters may have changed (load_state_dict, .to(), in-place edits are NOT tracked -- call
``invalidate()`` after editing parameters by hand).
"""
from __future__ import annotations

import contextlib
import ctypes
import math
from typing import Dict, Optional

import torch
from torch import nn

from . import _lib
from .encodings import coords_constant
from .spec import Entry, UNetGeometry, unet_entries

_FIR_DOWN = (0.125, 0.375, 0.375, 0.125)
_FIR_UP = (0.25, 0.75, 0.75, 0.25)


def _default_init(e: Entry, g: UNetGeometry) -> torch.Tensor:
    """Same distributions as a freshly constructed reference network (PyTorch defaults +
    zero_out, /root/reference/models/ops.py:9-11)."""
    from .synthetic import fourier_tables

    H, W = g.resolution
    r = e.role
    if r in ("conv_w", "linear_w"):
        fan_in = int(torch.tensor(e.shape[1:]).prod())
        bound = 1.0 / math.sqrt(fan_in)
        return torch.empty(e.shape).uniform_(-bound, bound)
    if r == "bias":
        return torch.empty(e.shape).uniform_(-0.05, 0.05)
    if r in ("conv_w_zero", "linear_w_zero", "bias_zero", "gn_b", "fourier_phase"):
        return torch.zeros(e.shape)
    if r == "gn_w":
        return torch.ones(e.shape)
    if r == "inv_sqrt2":
        return torch.tensor(1 / math.sqrt(2)).float()
    if r == "fir_down":
        return torch.tensor(_FIR_DOWN)
    if r == "fir_up":
        return torch.tensor(_FIR_UP)
    if r == "coords":
        # models/encoding.py:80-89
        phi = (0.5 - torch.arange(H) / H) * torch.pi
        theta = (1 - torch.arange(W) / W) * 2 * torch.pi - torch.pi
        phi, theta = torch.meshgrid([phi, theta], indexing="ij")
        return torch.stack([phi, theta])[None]
    if r == "fourier_freqs":
        return fourier_tables(H, W)[0]
    raise KeyError(r)


class _Engine:
    """Owns one r2dm_handle plus the torch tensors (blob, workspace) it points into."""

    def __init__(self, g: UNetGeometry, max_batch: int):
        L = _lib.lib()
        cfg = _lib.Config(
            in_channels=g.in_channels, out_channels=g.out_channels, height=g.resolution[0], width=g.resolution[1],
            base_channels=g.base_channels, temb_channels=g.temb_channels,
            channel_multiplier=(ctypes.c_int32 * 4)(*g.channel_multiplier),
            num_residual_blocks=(ctypes.c_int32 * 4)(*g.num_residual_blocks),
            gn_num_groups=g.gn_num_groups, gn_eps=g.gn_eps, attn_num_heads=g.attn_num_heads,
            coord_channels=g.coord_channels, max_batch=max_batch)
        h = ctypes.c_void_p()
        _lib.check(L.r2dm_create(ctypes.byref(h), ctypes.byref(cfg)))
        self.h = h
        self.max_batch = max_batch
        self.out_channels = g.out_channels
        self.blob: Optional[torch.Tensor] = None
        self.workspace: Optional[torch.Tensor] = None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _lib.lib().r2dm_destroy(self.h)
        except Exception:
            pass

    # -- weights -----------------------------------------------------------------------------
    def slots(self):
        L = _lib.lib()
        info = _lib.TensorInfo()
        for i in range(L.r2dm_num_tensors(self.h)):
            _lib.check(L.r2dm_tensor_at(self.h, i, ctypes.byref(info)))
            yield i, info.key.decode(), int(info.numel)

    def blob_bytes(self) -> int:
        return int(_lib.lib().r2dm_blob_bytes(self.h))

    def layout_hash(self) -> int:
        return int(_lib.lib().r2dm_blob_layout_hash(self.h))

    def bind(self, blob: torch.Tensor):
        _lib.require_gpu(blob, "weight blob")
        assert blob.dtype == torch.uint8 and blob.is_contiguous()
        _lib.check(_lib.lib().r2dm_bind_blob(self.h, blob.data_ptr(), blob.numel()))
        self.blob = blob

    def load(self, tensors: Dict[str, torch.Tensor], device: torch.device):
        """Pack `tensors` (U-Net state-dict keys + the two '__' constants) into a fresh blob."""
        L = _lib.lib()
        with torch.cuda.device(device):
            self.bind(torch.zeros(self.blob_bytes(), dtype=torch.uint8, device=device))
            st = _lib.stream_ptr(device)
            keep = []
            for i, key, numel in self.slots():
                if key not in tensors:
                    raise _lib.R2DMError(f"missing tensor {key!r} for the HIP engine")
                t = _lib.f32c(tensors[key]).to(device)
                keep.append(t)
                _lib.check(L.r2dm_load_tensor(self.h, i, t.data_ptr(), t.numel(), st))
            torch.cuda.current_stream(device).synchronize()  # sources may be freed after this

    # -- forward -----------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, cond: torch.Tensor) -> torch.Tensor:
        L = _lib.lib()
        B = x.shape[0]
        dev = x.device
        need = int(L.r2dm_workspace_bytes(self.h, B))
        if self.workspace is None or self.workspace.numel() < need or self.workspace.device != dev:
            self.workspace = torch.empty(need, dtype=torch.uint8, device=dev)
        out = torch.empty(B, self.out_channels, *x.shape[2:], device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(L.r2dm_unet_forward(self.h, x.data_ptr(), cond.data_ptr(), out.data_ptr(), B,
                                           self.workspace.data_ptr(), self.workspace.numel(), _lib.stream_ptr(dev)))
        return out


class EfficientUNet(nn.Module):
    def __init__(
        self,
        in_channels: int,
        resolution,
        out_channels: Optional[int] = None,
        base_channels: int = 128,
        temb_channels: Optional[int] = None,
        channel_multiplier=(1, 2, 4, 8),
        num_residual_blocks=(3, 3, 3, 3),
        gn_num_groups: int = 32 // 4,
        gn_eps: float = 1e-6,
        attn_num_heads: int = 8,
        coords_encoding: Optional[str] = "spherical_harmonics",
        ring: bool = True,
        max_batch: int = 8,
    ):
        super().__init__()
        if not ring:
            raise NotImplementedError("ring=False: the LiDAR models are always built with ring=True "
                                      "(/root/reference/utils/inference.py:50)")
        self.geometry = UNetGeometry.make(
            in_channels=in_channels, resolution=resolution, out_channels=out_channels, base_channels=base_channels,
            temb_channels=temb_channels, channel_multiplier=channel_multiplier,
            num_residual_blocks=num_residual_blocks, gn_num_groups=gn_num_groups, gn_eps=gn_eps,
            attn_num_heads=attn_num_heads, coords_encoding=coords_encoding)
        self.geometry.coord_channels  # raises ValueError for an unknown encoding
        self.resolution = self.geometry.resolution
        self.in_channels = in_channels
        self.out_channels = self.geometry.out_channels
        self.max_batch = max_batch
        for e in unet_entries(self.geometry):
            self._register(e, _default_init(e, self.geometry))
        self._engine: Optional[_Engine] = None
        self._packed_for = None
        self.precision = "fp32"
        # fp16 range guard of the default operand split: False (default) = a tripped guard switches this model to the wide-range split
        # ("fp32-bf16x3": exact 24-bit operands, fp32 operand range, ~1.6x slower) with ONE warning and the forward is repeated --
        # the reference samples any finite checkpoint (models/efficient_unet.py:269-295), so does the drop-in; True = raise R2DMRangeError
        self.strict_range = False
        self.range_fallbacks = 0  # how often that happened (0 or 1 per model: the switch is permanent)
        self._defer_range_check = False
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate())

    # -- generic parameter tree ----------------------------------------------------------------
    def _register(self, e: Entry, value: torch.Tensor):
        mod: nn.Module = self
        *path, leaf = e.key.split(".")
        for name in path:
            if not hasattr(mod, name):
                mod.add_module(name, nn.Module())
            mod = getattr(mod, name)
        if e.kind == "param":
            mod.register_parameter(leaf, nn.Parameter(value.float()))
        else:
            mod.register_buffer(leaf, value.float())

    def invalidate(self):
        """Forget the packed weights; they are rebuilt on the next forward."""
        self._packed_for = None

    def _apply(self, fn, *args, **kwargs):
        self.invalidate()
        return super()._apply(fn, *args, **kwargs)

    # -- constants the engine wants precomputed -------------------------------------------------
    @torch.no_grad()
    def _constants(self, sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        g = self.geometry
        out = {}
        if g.coords_encoding is not None:  # constant across steps and batch: evaluated once, on the host (encodings.py)
            out["__cenc"] = coords_constant(g.coords_encoding, sd["coords"], sd.get("coords_encoding.freqs"),
                                            sd.get("coords_encoding.phase"))
        half = g.base_channels // 2
        # models/ops.py:22-23 (host float32, exactly the reference expression)
        hcoef = -math.log(10_000) / (half - 1)
        out["__sin_freqs"] = torch.exp(hcoef * torch.arange(half))
        return out

    def _check_fixed_buffers(self, sd: Dict[str, torch.Tensor]):
        for k, v in sd.items():
            want = _FIR_DOWN if k.endswith("downsample.1.kernel") else _FIR_UP if k.endswith("upsample.0.kernel") else None
            if want is not None and not torch.allclose(v.detach().float().cpu(), torch.tensor(want)):
                raise _lib.R2DMError(f"{k}={v.tolist()}: the HIP resamplers implement the fixed [1,3,3,1] window only")

    def _ensure_packed(self, device: torch.device):
        key = (device.type, device.index)
        if self._engine is not None and self._packed_for == key:
            return
        if self._engine is None:
            self._engine = _Engine(self.geometry, self.max_batch)
            self.set_precision(self.precision)
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        self._check_fixed_buffers(sd)
        sd.update(self._constants(sd))
        self._engine.load(sd, device)
        self._packed_for = key

    # -- weight sharing across GPUs (one RCCL broadcast instead of N loads) ----------------------
    def packed_weights(self, device=None) -> torch.Tensor:
        """The engine's packed weight blob (uint8 tensor on the GPU), building it if necessary."""
        device = torch.device(device) if device is not None else self.coords.device
        if device.type != "cuda":
            raise _lib.R2DMError(f"packed weights live on the GPU; got device {device}")
        self._ensure_packed(device)
        return self._engine.blob

    def packed_layout_hash(self) -> int:
        """Fingerprint of the blob layout THIS model's engine plans (r2dm_blob_layout_hash): the packings depend on ``max_batch``, the
        device's CU count and experiment switches, not only on the configuration -- a blob may only be adopted between equal layouts."""
        if self._engine is None:
            self._engine = _Engine(self.geometry, self.max_batch)
            self.set_precision(self.precision)
        return self._engine.layout_hash()

    def adopt_packed_weights(self, blob: torch.Tensor, layout_hash: Optional[int] = None):
        """Bind a blob produced by ``packed_weights()`` of an identically configured model
        (e.g. received through ``torch.distributed.broadcast``) without loading a state dict.  ``layout_hash``: the packing
        model's ``packed_layout_hash()`` -- checked against this model's (the byte count alone cannot tell a blob
        planned for another batch size or tile choice, whose convolutions would then be silently wrong)."""
        if self._engine is None:
            self._engine = _Engine(self.geometry, self.max_batch)
            self.set_precision(self.precision)
        if layout_hash is None and blob.numel() != self._engine.blob_bytes():
            raise _lib.R2DMError(f"blob has {blob.numel()} bytes, engine expects {self._engine.blob_bytes()}")
        if layout_hash is not None and int(layout_hash) != self._engine.layout_hash():
            raise _lib.R2DMError(f"blob layout {int(layout_hash):#x} does not match this model's {self._engine.layout_hash():#x}: it was packed for another "
                                 "max_batch / device / tile selection (the packings of the convolutions differ); pack it with the same setup_model arguments")
        if blob.numel() != self._engine.blob_bytes():
            raise _lib.R2DMError(f"blob has {blob.numel()} bytes, engine expects {self._engine.blob_bytes()}")
        self._engine.bind(blob)
        self._packed_for = (blob.device.type, blob.device.index)

    def packed_weight_bytes(self) -> int:
        if self._engine is None:
            self._engine = _Engine(self.geometry, self.max_batch)
        return self._engine.blob_bytes()

    # -- arithmetic of the convolutions / attention on the matrix pipe --------------------------------------
    PRECISIONS = {"fp32": 2, "fp32-bf16x3": 3, "fp16": 1}
    _DEPRECATED_PRECISIONS = {"bf16x2": "fp32"}  # round 1's two-piece bf16 mode: the default is now both faster and more accurate

    def set_precision(self, precision: str = "fp32"):
        """``"fp32"`` (default) and ``"fp32-bf16x3"`` are PARITY modes: fp32 tensors, fp32 accumulation, products computed
        to fp32 accuracy or better; they differ in how the matrix pipe gets there.
        ``"fp32"``: every 3x3 / 1x1 convolution and the attention core split each fp32 operand to 22 bits into an fp16
        piece and a 2^11-scaled fp16 residual -- three fp16 MFMA products (the residual x residual term, 2^-22 relative, is
        dropped), two fp32 accumulators (conv_f16x2.hip, proj_f16x2.hip, attention.hip): fp32-class by measurement (error
        vs fp64 below an fp32 FMA chain's).  Operands pass through fp16, so |operand| < 65504 is required; weights are
        pre-scaled per layer by a power of two, activations are guarded by their producers' OBSERVED maxima, and
        ``check_range`` raises if a forward could have saturated.
        ``"fp32-bf16x3"``: three bf16 pieces / six products (exact 24-bit operands, fp32 operand range; the round-1 kernels,
        about 1.6x slower).
        ``"fp16"`` is the REDUCED-PRECISION bulk mode, the counterpart of the reference's fp16 autocast sampler
        (/root/reference/sample_and_save.py:70, utils/option.py:49): the same kernels with the fp16 piece alone -- one
        product per MAC (11-bit operands), fp32 accumulation, fp32 tensors, GroupNorm / softmax / the posterior update in
        fp32.  It has its own tolerance class (tests/test_hip_fp16_mode.py) and is never the default."""
        if precision in self._DEPRECATED_PRECISIONS:
            import warnings

            new = self._DEPRECATED_PRECISIONS[precision]
            warnings.warn(f"precision={precision!r} is deprecated (round-1 mode, superseded); using {new!r}", DeprecationWarning, stacklevel=2)
            precision = new
        if precision not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(self.PRECISIONS)}, got {precision!r}")
        self.precision = precision
        if self._engine is not None:
            _lib.check(_lib.lib().r2dm_set_conv_pieces(self._engine.h, self.PRECISIONS[precision]))
        return self

    def check_range(self):
        """Raise R2DMRangeError if an f16x2 convolution since the last check may have seen operands outside the fp16 range
        (synchronises the sampling stream).  Called after every stand-alone forward and at the end of a sampling loop."""
        if self._engine is not None and self._engine.blob is not None:
            dev = self._engine.blob.device
            with torch.cuda.device(dev):
                try:
                    _lib.check(_lib.lib().r2dm_check_range(self._engine.h, _lib.stream_ptr(dev)))
                except _lib.R2DMError as e:
                    if "may be outside the fp16 range" in str(e):  # (not the packer's "weight is not finite" flag: no split can run that)
                        raise _lib.R2DMRangeError(str(e)) from None
                    raise

    def range_report(self):
        """``[(site, bound)]`` of the forwards since the previous check, as the LAST ``check_range`` read them: one entry per guarded producer
        of a forward in walk order -- a GroupNorm's output bound ``|a| M + |d|`` or the recorded ``max|output|`` of a tensor the next
        fp16-operand kernel reads raw.  ``bound / 65504`` is the fraction of the fp16 range a layer used; >= 1 is a trip
        (``python -m r2dm_amd.check``)."""
        if self._engine is None:
            return []
        n = ctypes.c_int32(0)
        L = _lib.lib()
        _lib.check(L.r2dm_range_sites(self._engine.h, None, 0, ctypes.byref(n)))
        buf = (ctypes.c_float * max(n.value, 1))()
        _lib.check(L.r2dm_range_sites(self._engine.h, buf, n.value, ctypes.byref(n)))
        return [(L.r2dm_range_site_name(self._engine.h, k).decode(), float(buf[k])) for k in range(n.value) if k > 0 or buf[k] > 0]

    def check_range_or_fall_back(self) -> bool:
        """``check_range``; if the guard tripped and ``strict_range`` is off, switch this model to ``"fp32-bf16x3"`` for good
        (one warning) and return True: the caller repeats what it ran since the last check."""
        try:
            self.check_range()
            return False
        except _lib.R2DMRangeError as e:
            if self.strict_range or self.precision == "fp32-bf16x3":
                raise
            import warnings

            warnings.warn(f"r2dm_amd: the fp16 operand range guard tripped in precision={self.precision!r} ({e}); this checkpoint now runs on "
                          "the wide-range operand split 'fp32-bf16x3' (same parity class, about 1.6x slower). "
                          "Pass precision='fp32-bf16x3' to setup_model to start there, or strict_range=True to make this an error.",
                          RuntimeWarning, stacklevel=3)
            self.set_precision("fp32-bf16x3")
            self.range_fallbacks += 1
            return True

    @contextlib.contextmanager
    def deferred_range_check(self):
        """Inside: forwards do not synchronise for the range check; it runs once on exit (sampling loops)."""
        outer, self._defer_range_check = self._defer_range_check, True
        try:
            yield self
        except BaseException:
            self._defer_range_check = outer  # (an exception from the loop is not masked by a range error raised on top of it)
            raise
        else:
            self._defer_range_check = outer
            # a trip that is only seen here used to raise with strict_range off and leave the model on the fp16 path,
            # so every later call failed the same way.  Now the model switches (check_range_or_fall_back) and says so: the loops that
            # can replay themselves (sample: diffusion._Replay; repaint: from the saved generator states) never get here tripped.
            if not outer and self.check_range_or_fall_back():
                raise _lib.R2DMRangeFallback(
                    "the fp16 operand range guard tripped inside this loop: its results are not valid.  The model now runs the wide-range "
                    "operand split 'fp32-bf16x3' -- repeat the call (same generators re-seeded) to get them")

    # -- measurement aid ---------------------------------------------------------------------------
    def profile_convs(self, on: bool):
        """Bracket every MFMA-convolution launch with HIP events on the sampling stream (bench.py)."""
        if self._engine is None:
            raise _lib.R2DMError("run a forward pass first")
        _lib.check(_lib.lib().r2dm_profile_enable(self._engine.h, int(on)))

    def read_conv_profile(self):
        """-> (kernel milliseconds, algorithmic flops, launches) since profile_convs(True); resets."""
        ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _lib.check(_lib.lib().r2dm_profile_read(self._engine.h, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n)))
        return ms.value, fl.value, n.value

    CONV_CLASSES = ("conv_f16x2_kernel", "conv_bf16x3_*", "1x1 / in / out convolutions")

    def read_conv_profile_classes(self):
        """-> [(kernel class, milliseconds, algorithmic flops, launches)] since profile_convs(True); resets."""
        ms, fl, n = (ctypes.c_double * 3)(), (ctypes.c_double * 3)(), (ctypes.c_int64 * 3)()
        _lib.check(_lib.lib().r2dm_profile_read_classes(self._engine.h, ms, fl, n))
        return [(self.CONV_CLASSES[i], ms[i], fl[i], n[i]) for i in range(3)]

    def event_pair_overhead(self):
        """-> (microseconds an event pair spans around nothing, around an empty kernel) on the current stream: what the per-launch
        figures of profile_convs() contain beyond the kernel's own duration (rocprofv3's figure)."""
        a, b = ctypes.c_double(), ctypes.c_double()
        dev = self._engine.blob.device
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().r2dm_profile_event_overhead(self._engine.h, _lib.stream_ptr(dev), ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    # -- the hot path ----------------------------------------------------------------------------
    @torch.compiler.disable  # one ctypes call into libr2dm_hip.so: nothing for a tracing compiler to see (runs eagerly under
    def forward(self, images: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:  # torch.compile, sample_and_save.py:45)
        _lib.require_gpu(images, "images")
        if images.dim() != 4 or images.shape[1] != self.in_channels or tuple(images.shape[2:]) != self.resolution:
            raise ValueError(f"expected (B,{self.in_channels},{self.resolution[0]},{self.resolution[1]}), "
                             f"got {tuple(images.shape)}")
        x = _lib.f32c(images)
        B = x.shape[0]
        if timesteps.dim() == 0:  # efficient_unet.py:273-274
            timesteps = timesteps[None].repeat_interleave(B, dim=0)
        cond = _lib.f32c(timesteps.to(x.device))  # efficient_unet.py:275 `timesteps.to(h)`
        if cond.shape != (B,):
            raise ValueError(f"timesteps must have shape ({B},), got {tuple(timesteps.shape)}")
        self._ensure_packed(x.device)
        out = self._engine.forward(x, cond)
        if not self._defer_range_check and self.check_range_or_fall_back():
            out = self._engine.forward(x, cond)  # (again, on the wide-range split)
            self.check_range()
        return out
