"""Reverse-process samplers with the reference's API surface.

Mirrors (names, arguments, draw order, error behaviour):
  /root/reference/models/diffusion/base.py:8-163            GaussianDiffusion
  /root/reference/models/diffusion/continuous_time.py:66-317 ContinuousTimeGaussianDiffusion
  /root/reference/models/diffusion/discrete_time.py:51-201   DiscreteTimeGaussianDiffusion

MI355X design, not a translation:
  * All per-step schedule scalars (log-SNR, alpha, sigma, c, ...) are evaluated once, on the
    host, with the same float32 torch ops the reference uses, and uploaded as one table; the
    step loop itself launches no scalar kernels and never synchronises.
  * The posterior update ``x_t, prediction, noise -> x_s`` is ONE fused HIP kernel
    (``r2dm_posterior_step``) that replays the reference's float32 operation order with FMA
    contraction off.
  * Noise stays in ``torch.randn(generator=...)`` so that seeds mean what they mean in the
    reference (base.py:71-94).

Training-side members (loss, timestep sampling) are out of scope.
"""
from __future__ import annotations

import contextlib
import math
from functools import partial
from typing import List, Literal, Optional

import os
import warnings

import torch
from torch import nn
from torch.special import expm1

from . import _lib

try:  # progress bars are optional
    from tqdm.auto import tqdm
except Exception:  # pragma: no cover
    def tqdm(it, **_):
        return it

_OBJECTIVES = {"eps": 0, "v": 1, "x_0": 2}
# posterior kernel modes (csrc/posterior.hip)
_M_CT_DDPM, _M_CT_DDIM, _M_DT_DDPM, _M_DT_DDIM, _M_DT_DDIM_NOISE = 0, 1, 2, 3, 4
_NCOEF = 8


# --------------------------------------------------------------------------------------
# log-SNR schedules (continuous_time.py:14-63).  Evaluated on the HOST in float32.
# --------------------------------------------------------------------------------------
def _range_guard(model):
    """The denoiser's deferred fp16-range check (EfficientUNet.deferred_range_check) around a sampling loop: one
    synchronising check at the end instead of one per step; a no-op for any other denoiser."""
    guard = getattr(model, "deferred_range_check", None)
    return guard() if callable(guard) else contextlib.nullcontext()


def _early_range_check(model, num_steps: int) -> bool:
    """One synchronising range check after the first denoiser call of a loop (one sync per ``sample`` call).  True = the guard
    tripped and the denoiser switched itself to the wide-range operand split (EfficientUNet.check_range_or_fall_back): the
    caller repeats that call.  With ``strict_range`` a tripped guard raises, and short loops (<= 8 steps) skip the early
    check -- the deferred one at the loop's end reports it."""
    model = getattr(model, "_orig_mod", model)  # (torch.compile wrapper)
    check = getattr(model, "check_range_or_fall_back", None)
    if not callable(check):
        return False
    if getattr(model, "strict_range", False) and num_steps <= 8:
        return False
    return check()


_CHECK_EVERY = 32  # steps between two range checks of a sampling loop: one stream synchronisation each, at most so many steps to replay


class _Replay:
    """Recovery from a range-guard trip at ANY step of a ``sample`` loop (missing #2; the reference samples any
    finite checkpoint, /root/reference/models/diffusion/continuous_time.py:246-257).  The loop keeps ``x`` of the last step whose
    check passed and the noise it has drawn since; after step 0 of a longer loop, every ``_CHECK_EVERY`` steps and at the end the
    denoiser's guard is read (EfficientUNet.check_range_or_fall_back); on a trip the denoiser has switched itself to the wide-range
    operand split (one RuntimeWarning) and the loop goes back to that ``x`` and runs the steps since on the RECORDED noise -- same
    draws, no exception, at most ``_CHECK_EVERY`` steps lost, once per model.  ``strict_range``: the check raises instead (after
    step 0 of loops longer than 8 steps, else at the loop's end).  Any other denoiser: nothing is checked or kept."""

    def __init__(self, model, num_steps: int, every: int = _CHECK_EVERY):
        model = getattr(model, "_orig_mod", model)  # (torch.compile wrapper)
        check = getattr(model, "check_range_or_fall_back", None)
        self.check = check if callable(check) else None
        #  Only modes whose operands pass through fp16 can trip and fall back: a model that already runs the wide-range split
        # ("fp32-bf16x3": its check cannot trip) keeps neither a checkpoint nor the noise of up to 32 steps (0.5 GB at 128 x 2048, batch 8)
        self.record = self.check is not None and getattr(model, "precision", "fp32") != "fp32-bf16x3"
        self.strict = bool(getattr(model, "strict_range", False))
        self.n, self.every = num_steps, every
        self.ck_i, self.ck_x, self.noise, self.replays = 0, None, [], 0

    def start(self, x):
        self.ck_x = x if self.record else None

    def noise_for(self, i: int, draw):
        """The noise of step i: what was drawn for it before a replay, else ``draw()`` (recorded while a guard is watching)."""
        k = i - self.ck_i
        if k < len(self.noise):
            return self.noise[k]
        nz = draw()
        if self.record:
            self.noise.append(nz)
        return nz

    def due(self, i: int) -> bool:
        if self.check is None:
            return False
        early = i == 0 and self.n > 8  # a checkpoint the fp16 operand path cannot run at all: found after one step
        if self.strict:
            return early  # (its loop-end check is the deferred guard's: raises there)
        return early or i == self.n - 1 or (i + 1) % self.every == 0

    def after_step(self, i: int, x):
        """-> (next step, its input): (i + 1, x), or the checkpoint if the guard tripped since it was taken."""
        if not self.due(i):
            return i + 1, x
        if self.check():
            if not self.record:  # (cannot happen: a mode without a fallback raises inside check())
                raise _lib.R2DMError("range guard tripped in a mode that keeps no replay state")
            self.replays += 1
            self.noise = self.noise[: i + 1 - self.ck_i]
            self.record = False  # the model now runs the wide-range split: no second trip -- the recording ends with this replay
            return self.ck_i, self.ck_x
        self.ck_i, self.ck_x, self.noise = i + 1, (x if self.record else None), []
        return i + 1, x


def _progress_bar(total: int, desc: str, enabled: bool):
    try:
        return tqdm(total=total, desc=desc, leave=False, disable=not enabled)
    except TypeError:  # (the stand-in above: no bar)
        return None


def _rng_snapshot(rng, dev):
    """States of whatever ``GaussianDiffusion.randn`` draws from (base.py:71-94 forms), to repeat a loop on the same draws."""
    if rng is None:
        return torch.cuda.get_rng_state(dev) if torch.device(dev).type == "cuda" else torch.get_rng_state()
    if isinstance(rng, torch.Generator):
        return rng.get_state()
    return [g.get_state() for g in rng]


def _rng_restore(snap, rng, dev):
    if rng is None:
        torch.cuda.set_rng_state(snap, dev) if torch.device(dev).type == "cuda" else torch.set_rng_state(snap)
    elif isinstance(rng, torch.Generator):
        rng.set_state(snap)
    else:
        for g, s in zip(rng, snap):
            g.set_state(s)


def _log(t: torch.Tensor, eps: float = 1e-20) -> torch.Tensor:
    return torch.log(t.clamp(min=eps))


def log_snr_linear(t):
    return -_log(expm1(1e-4 + 10 * (t**2)))


def log_snr_cosine(t, logsnr_min: float = -15, logsnr_max: float = 15):
    t_min = math.atan(math.exp(-0.5 * logsnr_max))
    t_max = math.atan(math.exp(-0.5 * logsnr_min))
    return -2 * _log(torch.tan(t_min + t * (t_max - t_min)))


def log_snr_cosine_shifted(t, image_d, noise_d, logsnr_min: float = -15, logsnr_max: float = 15):
    return log_snr_cosine(t, logsnr_min, logsnr_max) + 2 * math.log(noise_d / image_d)


def log_snr_cosine_interpolated(t, image_d, noise_d_low, noise_d_high, logsnr_min: float = -15,
                                logsnr_max: float = 15):
    low = log_snr_cosine_shifted(t, image_d, noise_d_low, logsnr_min, logsnr_max)
    high = log_snr_cosine_shifted(t, image_d, noise_d_high, logsnr_min, logsnr_max)
    return t * low + (1 - t) * high


def log_snr_to_alpha_sigma(log_snr):
    return log_snr.sigmoid().sqrt(), (-log_snr).sigmoid().sqrt()


def discrete_tables(num_training_steps: int, schedule: str):
    """beta / alpha_bar / alpha_bar_prev / snr in float64 -> float32 (discrete_time.py:12-78)."""
    T = num_training_steps
    if schedule == "linear":
        s = 1000 / T
        beta = torch.linspace(s * 0.0001, s * 0.02, T, dtype=torch.float64)
    elif schedule in ("cosine", "sigmoid"):
        t = torch.linspace(0, T, T + 1, dtype=torch.float64) / T
        if schedule == "cosine":
            ab = torch.cos((t + 0.008) / (1 + 0.008) * math.pi * 0.5) ** 2
        else:
            start, end, tau = -3, 3, 1
            v0, v1 = torch.tensor(start / tau).sigmoid(), torch.tensor(end / tau).sigmoid()
            ab = (-((t * (end - start) + start) / tau).sigmoid() + v1) / (v1 - v0)
        ab = ab / ab[0]
        beta = torch.clip(1 - (ab[1:] / ab[:-1]), 0, 0.999)
    else:
        raise ValueError(f"invalid beta schedule {schedule}")
    alpha_bar = torch.cumprod(1 - beta, dim=0)
    alpha_bar_prev = torch.cat([alpha_bar.new_ones(1), alpha_bar[:-1]])
    snr = alpha_bar / (1 - alpha_bar)
    return beta.float(), alpha_bar.float(), alpha_bar_prev.float(), snr.float()


# --------------------------------------------------------------------------------------
class GaussianDiffusion(nn.Module):
    """Common state and RNG plumbing (base.py:8-163)."""

    def __init__(
        self,
        model: nn.Module,
        sampling: Literal["ddpm", "ddim"] = "ddpm",
        prediction_type: Literal["eps", "v", "x_0"] = "eps",
        loss_type="l2",
        num_training_steps: Optional[int] = 1000,
        noise_schedule: str = "linear",
        min_snr_loss_weight: bool = True,
        min_snr_gamma: float = 5.0,
        sampling_resolution: Optional[tuple] = None,
        clip_sample: bool = True,
        clip_sample_range: float = 1,
    ):
        super().__init__()
        self.model = model
        self.sampling = sampling
        self.num_training_steps = num_training_steps
        self.objective = prediction_type
        self.noise_schedule = noise_schedule
        self.loss_type = loss_type
        self.min_snr_loss_weight = min_snr_loss_weight
        self.min_snr_gamma = min_snr_gamma
        self.clip_sample = clip_sample
        self.clip_sample_range = clip_sample_range
        if sampling_resolution is None:
            assert hasattr(self.model, "resolution")
            assert hasattr(self.model, "in_channels")
            self.sampling_shape = (self.model.in_channels, *self.model.resolution)
        else:
            assert len(sampling_resolution) == 2
            assert hasattr(self.model, "in_channels")
            self.sampling_shape = (self.model.in_channels, *sampling_resolution)
        self.setup_parameters()
        self.register_buffer("_dummy", torch.tensor([]))

    @property
    def device(self):
        return self._dummy.device

    # -- RNG: identical draw shapes / order to base.py:71-94 -------------------------------
    def randn(self, *shape, rng: List[torch.Generator] | torch.Generator | None = None, **kwargs):
        if rng is None:
            return torch.randn(*shape, **kwargs)
        elif isinstance(rng, torch.Generator):
            return torch.randn(*shape, generator=rng, **kwargs)
        elif isinstance(rng, list):
            assert len(rng) == shape[0]
            # (base.py:81-85 stacks per-sample draws; drawn straight into the rows of the result here -- the same values from the same
            # generator states, one copy kernel and 2 x the bytes less per step)
            if not rng:
                return torch.stack([])  # (the reference's error for an empty batch)
            out = torch.empty(*shape, **kwargs)
            for i, r in enumerate(rng):
                torch.randn(*shape[1:], generator=r, out=out[i])
            return out
        else:
            raise ValueError(f"invalid rng: {rng}")

    def randn_like(self, x, rng=None):
        return self.randn(*x.shape, rng=rng, device=x.device, dtype=x.dtype)

    def _randn_like_ahead(self, x, rng=None):
        """``(noise, event)``: the step's noise drawn on a side stream BEFORE the denoiser call, so that its small kernels (one
        ``torch.randn`` per sample generator plus the stack: nine launches, ~45 us per step in series) fill the gaps of the
        denoiser's stream instead of following it.  Same draws in the same order as base.py:71-94 (the denoiser draws nothing;
        a generator's state advances on the host at launch time); the caller waits for ``event`` before it reads ``noise``.
        ``R2DM_NOISE_STREAM=0`` or a CPU tensor: a plain ``randn_like`` and no event."""
        if os.environ.get("R2DM_DEBUG_FIXED_NOISE") == "1":  # timing experiment only (WRONG samples): what do the step's noise launches cost?
            if not self.__dict__.get("_fixed_noise_warned"):
                self.__dict__["_fixed_noise_warned"] = True
                warnings.warn("R2DM_DEBUG_FIXED_NOISE=1: every step reuses ONE noise tensor -- the samples are WRONG (timing experiment only)", RuntimeWarning)
            z = self.__dict__.get("_fixed_noise")
            if z is None or z.shape != x.shape or z.device != x.device:
                z = self.__dict__["_fixed_noise"] = self.randn_like(x, rng=rng)
            return z, None
        if x.device.type != "cuda" or os.environ.get("R2DM_NOISE_STREAM", "1") == "0":
            return self.randn_like(x, rng=rng), None
        side = self.__dict__.get("_noise_stream")
        if side is None or side.device != x.device:
            side = self.__dict__["_noise_stream"] = torch.cuda.Stream(device=x.device)
        main = torch.cuda.current_stream(x.device)
        with torch.cuda.stream(side):
            noise = self.randn_like(x, rng=rng)
            event = torch.cuda.Event()
            event.record(side)
        noise.record_stream(main)
        return noise, event

    def setup_parameters(self) -> None:
        raise NotImplementedError

    # Where the schedule scalars are evaluated.  "host" (default): float32 on the CPU, row by row -- bit-identical
    # to the reference's CPU run (tests/golden/schedule.npz) on any machine.  "device": with the same torch ops on the tensors' own device,
    # as the reference does when it itself runs on a GPU (continuous_time.py:203-206,248-249: `linspace(..., device=self.device)`, log_snr,
    # alpha, sigma of device tensors).  The two differ in the last bits of libm; that only matters where a scalar is a rounding residue --
    # DDIM with eta = 1, whose c_2 = sqrt(1 - alpha_s^2 - c_1^2) is one on the first and last step (profiles/r05_fuzz.txt, case 15).
    schedule_on: str = "host"

    def set_schedule_on(self, where: str):
        if where not in ("host", "device"):
            raise ValueError(f"schedule_on must be 'host' or 'device', got {where!r}")
        self.schedule_on = where
        self.__dict__.pop("_tables", None)
        return self

    def _on_device(self, dev) -> bool:
        return self.schedule_on == "device" and torch.device(dev).type == "cuda"

    # -- training side: out of scope ------------------------------------------------------
    def forward(self, *args, **kwargs):
        raise NotImplementedError(
            "r2dm_amd builds the sampling path only; the training loss "
            "(reference base.py:122-149) is out of scope -- see DESIGN.md")

    p_loss = forward

    def _objective_id(self) -> int:
        if self.objective not in _OBJECTIVES:
            raise ValueError(f"invalid objective {self.objective}")
        return _OBJECTIVES[self.objective]

    def _clip(self) -> float:
        return float(self.clip_sample_range) if self.clip_sample else -1.0

    def _posterior(self, x_t, pred, noise, coef, mode_id: int):
        """x_s = posterior(x_t, prediction, noise) in one HIP launch."""
        return _lib.posterior_step(x_t, pred, noise, coef, mode_id, self._objective_id(), self._clip())


# --------------------------------------------------------------------------------------
class ContinuousTimeGaussianDiffusion(GaussianDiffusion):
    """Variational-diffusion-style continuous-time process (continuous_time.py:66-317)."""

    def __init__(
        self,
        model: nn.Module,
        prediction_type: Literal["eps", "v", "x_0"] = "eps",
        loss_type="l2",
        noise_schedule: Literal["linear", "cosine", "cosine_shifted", "cosine_interpolated"] = "cosine",
        min_snr_loss_weight: bool = True,
        min_snr_gamma: float = 5.0,
        sampling_resolution: Optional[tuple] = None,
        clip_sample: bool = True,
        clip_sample_range: float = 1,
        image_d: float = None,
        noise_d_low: float = None,
        noise_d_high: float = None,
    ):
        self.image_d, self.noise_d_low, self.noise_d_high = image_d, noise_d_low, noise_d_high
        super().__init__(
            model=model,
            sampling="ddpm",
            prediction_type=prediction_type,
            loss_type=loss_type,
            num_training_steps=None,
            noise_schedule=noise_schedule,
            min_snr_loss_weight=min_snr_loss_weight,
            min_snr_gamma=min_snr_gamma,
            sampling_resolution=sampling_resolution,
            clip_sample=clip_sample,
            clip_sample_range=clip_sample_range,
        )

    def setup_parameters(self) -> None:
        if self.noise_schedule == "linear":
            f = log_snr_linear
        elif self.noise_schedule == "cosine":
            f = log_snr_cosine
        elif self.noise_schedule == "cosine_shifted":
            assert self.image_d is not None and self.noise_d_low is not None
            f = partial(log_snr_cosine_shifted, image_d=self.image_d, noise_d=self.noise_d_low)
        elif self.noise_schedule == "cosine_interpolated":
            assert self.image_d is not None and self.noise_d_low is not None and self.noise_d_high is not None
            f = partial(log_snr_cosine_interpolated, image_d=self.image_d, noise_d_low=self.noise_d_low,
                        noise_d_high=self.noise_d_high)
        else:
            raise ValueError(f"invalid beta schedule: {self.noise_schedule}")
        self._log_snr_1d = f

    def log_snr(self, t: torch.Tensor) -> torch.Tensor:
        """(B,) -> (B,1,1,1), as continuous_time.py:14-29."""
        return self._log_snr_1d(t)[:, None, None, None]

    def get_network_condition(self, steps):
        return self._log_snr_1d(steps)

    # -- host-side coefficient table -------------------------------------------------------
    def _coefficient_row(self, t: torch.Tensor, s: torch.Tensor, mode: str, ddim_eta: float):
        """Scalars of continuous_time.py:203-229 for ONE (t, s) pair, as 1-element float32 tensors."""
        lt, coef = self._coefficient_block(t, s, mode, ddim_eta)
        return lt, coef[0]

    def _coefficient_block(self, t: torch.Tensor, s: torch.Tensor, mode: str, ddim_eta: float):
        """Scalars of continuous_time.py:203-229 for (N,) step pairs, with the reference's own torch expressions, WHERE t lives:
        -> (log-SNR (N,), coefficients (N, 8))."""
        lt, ls = self._log_snr_1d(t), self._log_snr_1d(s)
        a_t, s_t = log_snr_to_alpha_sigma(lt)
        a_s, s_s = log_snr_to_alpha_sigma(ls)
        z = torch.zeros_like(lt)
        if mode == "ddpm":
            c = -expm1(lt - ls)
            return lt, torch.stack([a_t, s_t, a_s, s_s, c, s_s * c.sqrt(), z, z], dim=-1)
        if mode == "ddim":
            c_1 = ddim_eta * s_s / s_t * (1 - a_t**2 / a_s**2).sqrt()
            c_2 = (1 - a_s**2 - c_1**2).sqrt()
            return lt, torch.stack([a_t, s_t, a_s, s_s, z, z, c_1, c_2], dim=-1)
        raise ValueError(f"invalid mode {mode}")

    def _coefficients(self, step_t: torch.Tensor, step_s: torch.Tensor, mode: str, ddim_eta: float):
        """Per-row schedule scalars, float32 on the host.  Returns (cond (N,), coef (N,8), kernel mode).

        Each row is evaluated on 1-element tensors: torch's CPU elementwise kernels round the last bit
        differently in their SIMD body and in their scalar tail, and the reference evaluates the schedule on
        (B,)-shaped tensors (continuous_time.py:203-206), i.e. on the scalar path for its CPU-runnable
        batch sizes (BASELINE configs[0]: batch 1).  Row-wise evaluation reproduces those values bit for
        bit and makes the table independent of the number of steps."""
        if mode not in ("ddpm", "ddim"):
            raise ValueError(f"invalid mode {mode}")
        if self._on_device(step_t.device):  # (B,)-shaped device tensors through the same ops, as the reference on a GPU
            lt, coef = self._coefficient_block(step_t.detach().float().reshape(-1), step_s.detach().to(step_t.device).float().reshape(-1), mode, ddim_eta)
            return lt, coef.contiguous(), (_M_CT_DDPM if mode == "ddpm" else _M_CT_DDIM)
        step_t = step_t.detach().to("cpu", torch.float32).reshape(-1)
        step_s = step_s.detach().to("cpu", torch.float32).reshape(-1)
        memo, conds, rows = {}, [], []
        for i in range(step_t.numel()):
            key = (step_t[i].item(), step_s[i].item())
            if key not in memo:
                memo[key] = self._coefficient_row(step_t[i:i + 1], step_s[i:i + 1], mode, ddim_eta)
            conds.append(memo[key][0])
            rows.append(memo[key][1])
        return torch.cat(conds), torch.stack(rows).contiguous(), (_M_CT_DDPM if mode == "ddpm" else _M_CT_DDIM)

    def _sample_tables(self, num_steps: int, batch_size: int, mode: str, ddim_eta: float, dev):
        """(cond (S, B), coef (S, B, 8), kernel mode) of a whole ``sample`` call on the device: ``_table_rows`` walked to its
        end (ONE implementation builds, caches and evicts the tables -- the one ``sample`` uses)."""
        row, mode_id = self._table_rows(num_steps, batch_size, mode, ddim_eta, dev)
        for i in range(num_steps):
            row(i)
        cond, coef, _ = self.__dict__["_tables"][self._table_key(num_steps, batch_size, mode, ddim_eta, dev)]
        return cond, coef, mode_id

    def _table_key(self, num_steps, batch_size, mode, ddim_eta, dev):
        return (num_steps, batch_size, mode, float(ddim_eta), str(torch.device(dev)), self.noise_schedule, self.image_d, self.noise_d_low, self.noise_d_high,
                self.schedule_on)

    def _table_rows(self, num_steps: int, batch_size: int, mode: str, ddim_eta: float, dev):
        """``row(i) -> (cond (B,), coef (B, 8))`` and ``mode_id`` for a ``sample`` call.  A cached table is indexed; a new one is
        built ROW BY ROW while the loop runs: each row is ~0.3 ms of host work (about thirty 1-element torch ops), 80 ms for 256
        steps -- with the whole table built up front that was 4.5 % of a cold 256-step call with the GPU idle (round 3:
        scripts/step_times.py, profiles/r03_step_times.txt), whereas one row per step hides under the ~6 ms the GPU needs per
        step.  Rows travel through pinned host memory (an asynchronous copy on the sampling stream: the host never waits for the
        GPU) into the device table, which enters the cache when its last row is in."""
        dev = torch.device(dev)
        key = self._table_key(num_steps, batch_size, mode, ddim_eta, dev)
        cache = self.__dict__.setdefault("_tables", {})
        hit = cache.get(key)
        if hit is not None:
            cond, coef, mode_id = hit
            return (lambda i: (cond[i], coef[i])), mode_id
        if mode not in ("ddpm", "ddim"):
            raise ValueError(f"invalid mode {mode}")
        mode_id = _M_CT_DDPM if mode == "ddpm" else _M_CT_DDIM
        if self._on_device(dev):
            # the whole table at once, on the device: `steps` is the reference's own device linspace (continuous_time.py:248), and a device
            # elementwise kernel computes every element by the same instructions whatever the tensor's shape -- (S,) here, (B,) per step there
            steps = torch.linspace(1.0, 0.0, num_steps + 1, device=dev)
            lt, k = self._coefficient_block(steps[:-1], steps[1:], mode, ddim_eta)
            cond = lt[:, None].repeat_interleave(batch_size, dim=1).contiguous()
            coef = k[:, None, :].repeat_interleave(batch_size, dim=1).contiguous()
            if len(cache) >= 8:
                cache.pop(next(iter(cache)))
            cache[key] = (cond, coef, mode_id)
            return (lambda i: (cond[i], coef[i])), mode_id
        steps = torch.linspace(1.0, 0.0, num_steps + 1)
        pin = dev.type == "cuda"
        h_cond = torch.empty(num_steps, batch_size, pin_memory=pin)
        h_coef = torch.empty(num_steps, batch_size, _NCOEF, pin_memory=pin)
        cond = torch.empty(num_steps, batch_size, device=dev)
        coef = torch.empty(num_steps, batch_size, _NCOEF, device=dev)

        def row(i):
            c, k, _ = self._coefficients(steps[i:i + 1], steps[i + 1:i + 2], mode, ddim_eta)
            h_cond[i] = c
            h_coef[i] = k[0]
            cond[i].copy_(h_cond[i], non_blocking=True)
            coef[i].copy_(h_coef[i], non_blocking=True)
            if i == num_steps - 1:
                if len(cache) >= 8:
                    cache.pop(next(iter(cache)))
                cache[key] = (cond, coef, mode_id)
            return cond[i], coef[i]

        return row, mode_id

    @torch.inference_mode()
    def p_step(self, x_t, step_t, step_s, rng=None, mode: Literal["ddpm", "ddim"] = "ddpm",
               ddim_eta: float = 0.0, _early_check: int = 0):
        """One reverse step p(z_s | z_t), 0 <= s < t <= 1 (continuous_time.py:192-232).
        (``_early_check``: internal -- the first step of a loop under the deferred range guard passes its step count.)"""
        self._objective_id()
        dev = x_t.device
        if self._on_device(dev):
            step_t, step_s = step_t.to(dev), step_s.to(dev)
        cond, coef, mode_id = self._coefficients(step_t, step_s, mode, ddim_eta)
        prediction = self.model(x_t, cond.to(dev))
        if _early_check and _early_range_check(self.model, _early_check):
            prediction = self.model(x_t, cond.to(dev))  # (repeated on the wide-range operand split)
        noise = self.randn_like(x_t, rng=rng)
        return self._posterior(x_t, prediction, noise, coef.to(dev), mode_id)

    @torch.inference_mode()
    def sample(self, batch_size: int, num_steps: int, progress: bool = True, rng=None,
               return_all: bool = False, mode: Literal["ddpm", "ddim"] = "ddpm", ddim_eta: float = 0.0):
        """Ancestral / DDIM sampling from t=1 to t=0 (continuous_time.py:234-258)."""
        dev = self.device
        self._objective_id()
        x = self.randn(batch_size, *self.sampling_shape, rng=rng, device=dev)
        if return_all:
            out = [x]
        row, mode_id = self._table_rows(num_steps, batch_size, mode, ddim_eta, dev)
        rp = _Replay(self.model, num_steps)  # (a range-guard trip at any step: back to the last checked x, on the recorded noise)
        rp.start(x)
        bar = _progress_bar(num_steps, "sampling", progress)
        with _range_guard(self.model):
            i = 0
            while i < num_steps:
                cond_i, coef_i = row(i)
                noise, drawn = rp.noise_for(i, lambda: self._randn_like_ahead(x, rng=rng))
                prediction = self.model(x, cond_i)
                if drawn is not None:
                    torch.cuda.current_stream(x.device).wait_event(drawn)
                nxt, x = rp.after_step(i, self._posterior(x, prediction, noise, coef_i, mode_id))
                if return_all:
                    del out[nxt + 1:]  # (after a replay: the steps that are run again; else nothing)
                    if nxt > i:
                        out.append(x)
                if bar is not None:
                    bar.update(nxt - i)
                i = nxt
        if bar is not None:
            bar.close()
        return torch.stack(out) if return_all else x

    # -- forward process pieces used by RePaint (continuous_time.py:169-190) ----------------
    def _alpha_sigma_rows(self, steps: torch.Tensor) -> torch.Tensor:
        """(alpha, sigma) per row, float32 on the host, each row evaluated on 1-element tensors (see _coefficients)."""
        if self._on_device(steps.device):
            return torch.stack(log_snr_to_alpha_sigma(self._log_snr_1d(steps.detach().float().reshape(-1))), dim=-1).contiguous()
        steps = steps.detach().to("cpu", torch.float32).reshape(-1)
        rows = []
        for i in range(steps.numel()):
            a, sg = log_snr_to_alpha_sigma(self._log_snr_1d(steps[i:i + 1]))
            rows.append(torch.cat([a, sg]))
        return torch.stack(rows).contiguous()

    def _q_coef_rows(self, step_t: torch.Tensor, step_s: torch.Tensor) -> torch.Tensor:
        """(alpha_t/alpha_s, sqrt(sigma_t^2 - alpha_ts^2 sigma_s^2)) per row (continuous_time.py:181-189)."""
        if self._on_device(step_t.device):
            a_t, s_t = log_snr_to_alpha_sigma(self._log_snr_1d(step_t.detach().float().reshape(-1)))
            a_s, s_s = log_snr_to_alpha_sigma(self._log_snr_1d(step_s.detach().to(step_t.device).float().reshape(-1)))
            a_ts = a_t / a_s
            return torch.stack([a_ts, (s_t.pow(2) - a_ts.pow(2) * s_s.pow(2)).sqrt()], dim=-1).contiguous()
        step_t = step_t.detach().to("cpu", torch.float32).reshape(-1)
        step_s = step_s.detach().to("cpu", torch.float32).reshape(-1)
        rows = []
        for i in range(step_t.numel()):
            a_t, s_t = log_snr_to_alpha_sigma(self._log_snr_1d(step_t[i:i + 1]))
            a_s, s_s = log_snr_to_alpha_sigma(self._log_snr_1d(step_s[i:i + 1]))
            a_ts = a_t / a_s
            var = s_t.pow(2) - a_ts.pow(2) * s_s.pow(2)
            rows.append(torch.cat([a_ts, var.sqrt()]))
        return torch.stack(rows).contiguous()

    def q_step_from_x_0(self, x_0, step_t, rng=None):
        """Forward process q(z_t | x_0) (continuous_time.py:169-176); HIP kernel r2dm_q_step."""
        noise = self.randn_like(x_0, rng=rng)
        coef = self._alpha_sigma_rows(step_t).to(x_0.device)  # x_0 * alpha + sigma * noise
        return _lib.q_step(x_0, noise, coef), noise

    def q_step(self, x_s, step_t, step_s, rng=None):
        """q(z_t | z_s), 0 < s < t < 1 (continuous_time.py:178-190); HIP kernel r2dm_q_step."""
        coef = self._q_coef_rows(step_t, step_s).to(x_s.device)
        var_noise = self.randn_like(x_s, rng=rng)
        return _lib.q_step(x_s, var_noise, coef)

    @torch.inference_mode()
    def repaint(self, known, mask, num_steps: int, num_resample_steps: int = 1, jump_length: int = 1,
                progress: bool = True, rng=None, return_all: bool = False):
        """RePaint inpainting on top of the same p_step (continuous_time.py:260-317).  A range-guard trip that only shows after the
        first step (``_early_check``) ends the loop with the denoiser switched to the wide-range split: the call is then repeated
        once, from the generator states it started with -- same draws, no exception (the reference runs any finite checkpoint)."""
        snap = _rng_snapshot(rng, self.device)
        try:
            return self._repaint(known, mask, num_steps, num_resample_steps, jump_length, progress, rng, return_all)
        except _lib.R2DMRangeFallback:
            _rng_restore(snap, rng, self.device)
            return self._repaint(known, mask, num_steps, num_resample_steps, jump_length, progress, rng, return_all)

    def _repaint(self, known, mask, num_steps, num_resample_steps, jump_length, progress, rng, return_all):
        assert num_resample_steps > 0
        assert jump_length > 0
        B = known.shape[0]
        dev = self.device
        x_t = self.randn(B, *self.sampling_shape, rng=rng, device=dev)
        steps = torch.linspace(1, 0, num_steps + 1)[None].repeat_interleave(B, dim=0)
        if return_all:
            out = [x_t]
        with _range_guard(self.model):
            for i in tqdm(range(num_steps), desc="RePaint", leave=False, disable=not progress):
                for j in range(num_resample_steps):
                    t, s = steps[:, [i]], steps[:, [i + 1]]
                    r = t + torch.linspace(0, 1, jump_length + 1)[None] * (s - t)
                    x = x_t
                    for k in range(jump_length):
                        # q_step_from_x_0(known) and the mask blend are one kernel; the draw order (known-region noise,
                        # then p_step's noise) is the reference's
                        noise_k = self.randn_like(known, rng=rng)
                        first = i == 0 and j == 0 and k == 0
                        unknown_s = self.p_step(x, r[:, k], r[:, k + 1], rng=rng, _early_check=num_steps * num_resample_steps * jump_length if first else 0)
                        x = _lib.repaint_blend(known, noise_k, unknown_s, mask, self._alpha_sigma_rows(r[:, k + 1]).to(dev))
                    x_s = x
                    if return_all:
                        out.append(x_s)
                    if (i == num_steps - 1) or (j == num_resample_steps - 1):
                        x_t = x
                        break
                    for k in range(jump_length, 0, -1):
                        x = self.q_step(x, r[:, k - 1], r[:, k], rng=rng)
                    x_t = x
        return torch.stack(out) if return_all else x_s


# --------------------------------------------------------------------------------------
class DiscreteTimeGaussianDiffusion(GaussianDiffusion):
    """DDPM / DDIM over an integer-step beta schedule (discrete_time.py:51-201)."""

    def setup_parameters(self) -> None:
        assert self.num_training_steps is not None
        beta, ab, abp, snr = discrete_tables(self.num_training_steps, self.noise_schedule)
        v4 = lambda t: t[:, None, None, None]
        self.register_buffer("beta", v4(beta))
        self.register_buffer("alpha_bar", v4(ab))
        self.register_buffer("alpha_bar_prev", v4(abp))
        self.register_buffer("snr", v4(snr))

    def get_network_condition(self, steps):
        return steps

    def _coefficients(self, steps: torch.Tensor, mode: str, eta: float):
        """Row scalars of discrete_time.py:135-177 on the host (float32, same op order)."""
        idx = steps.detach().to("cpu", torch.long)
        beta = self.beta.detach().cpu()[idx, 0, 0, 0]
        ab = self.alpha_bar.detach().cpu()[idx, 0, 0, 0]
        abp = self.alpha_bar_prev.detach().cpu()[idx, 0, 0, 0]
        alpha = 1 - beta
        live = (idx != 0).float()  # var_noise[steps == 0] *= 0 (discrete_time.py:162,176)
        if self.objective == "eps":
            k0, k1 = ab.rsqrt(), (ab.reciprocal() - 1).sqrt()
        elif self.objective == "v":
            k0, k1 = ab.sqrt(), (1 - ab).sqrt()
        else:
            k0 = k1 = torch.zeros_like(ab)
        if mode == "ddpm":
            x0c = abp.sqrt() * beta / (1 - ab)
            xtc = (1 - abp) * alpha.sqrt() / (1 - ab)
            var = (beta * (1 - abp) / (1 - ab)).clamp(min=1e-20)
            sd = (0.5 * var.log()).exp() * live
            coef = torch.stack([k0, k1, x0c, xtc, sd, torch.zeros_like(ab), torch.zeros_like(ab),
                                torch.zeros_like(ab)], dim=-1)
            return coef.contiguous(), _M_DT_DDPM
        if mode == "ddim":
            var = (1 - abp) / (1 - ab) * (1 - ab / abp)
            std = eta * torch.sqrt(var)
            coef = torch.stack([k0, k1, ab.sqrt(), (1 - ab).sqrt(), (1 - abp - std**2).sqrt(), abp.sqrt(),
                                std * live, torch.zeros_like(ab)], dim=-1)
            return coef.contiguous(), (_M_DT_DDIM_NOISE if eta > 0 else _M_DT_DDIM)
        raise ValueError(f"invalid mode {mode}")

    @torch.inference_mode()
    def p_step(self, x_t, steps, rng=None, mode: Literal["ddpm", "ddim"] = "ddim", eta: float = 0.0):
        self._objective_id()
        coef, mode_id = self._coefficients(steps, mode, eta)
        prediction = self.model(x_t, steps.to(x_t.device))
        # draw order as the reference: DDPM always draws; DDIM only when eta > 0 (:160,:174)
        noise = self.randn_like(x_t, rng=rng) if mode_id != _M_DT_DDIM else None
        return self._posterior(x_t, prediction, noise, coef.to(x_t.device), mode_id)

    @torch.inference_mode()
    def sample(self, batch_size: int, num_steps: int, progress: bool = True, rng=None,
               return_all: bool = False, mode: Literal["ddpm", "ddim"] = "ddpm"):
        """t = num_steps-1 ... 0 without respacing (discrete_time.py:182-201)."""
        dev = self.device
        self._objective_id()
        x = self.randn(batch_size, *self.sampling_shape, rng=rng, device=dev)
        if return_all:
            out = [x]
        order = torch.arange(num_steps - 1, -1, -1)
        coef, mode_id = self._coefficients(order, mode, 0.0)
        coef = coef[:, None, :].expand(num_steps, batch_size, _NCOEF).contiguous().to(dev)
        cond = order[:, None].expand(num_steps, batch_size).contiguous().to(dev)
        rp = _Replay(self.model, num_steps)  # (as ContinuousTimeGaussianDiffusion.sample)
        rp.start(x)
        bar = _progress_bar(num_steps, "sampling", progress)
        with _range_guard(self.model):
            i = 0
            while i < num_steps:
                noise, drawn = rp.noise_for(i, lambda: self._randn_like_ahead(x, rng=rng)) if mode_id != _M_DT_DDIM else (None, None)
                prediction = self.model(x, cond[i])
                if drawn is not None:
                    torch.cuda.current_stream(x.device).wait_event(drawn)
                nxt, x = rp.after_step(i, self._posterior(x, prediction, noise, coef[i], mode_id))
                if return_all:
                    del out[nxt + 1:]
                    if nxt > i:
                        out.append(x)
                if bar is not None:
                    bar.update(nxt - i)
                i = nxt
        if bar is not None:
            bar.close()
        return torch.stack(out) if return_all else x
