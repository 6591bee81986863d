"""Test-side wrappers around the single-kernel C-ABI entry points of libr2dm_hip.so."""
import torch

from r2dm_amd import _lib


def _st(t):
    return _lib.stream_ptr(t.device)


def set_conv_pieces(n):
    """Operand split of the single-kernel conv entry: 2 = fp16 + scaled fp16 residual where the shape allows (default),
    3 = three bf16 pieces."""
    _lib.check(_lib.lib().r2dm_set_conv_pieces(None, n))


def conv2d_ring(x, w, b, aff=None, prologue=0, residual=None, scale=None, io16=0):
    """io16 (with R2DM_TEST_IO16 set to the same value by the caller): bit 0 -- x is a half tensor, bit 1 -- the residual is and y will be."""
    L = _lib.lib()
    B, cin, H, W = x.shape
    cout, k = w.shape[0], w.shape[-1]
    w, b = _lib.f32c(w), _lib.f32c(b)
    x = x.detach().half().contiguous() if io16 & 1 else _lib.f32c(x)
    if residual is not None:
        residual = residual.detach().half().contiguous() if io16 & 2 else _lib.f32c(residual)
    packed = torch.empty(L.r2dm_conv_packed_elems(cout, cin, k, B, H, W), device=x.device)
    y = torch.empty(B, cout, H, W, device=x.device, dtype=torch.float16 if io16 & 2 else torch.float32)
    sc = None if scale is None else torch.tensor([scale], device=x.device, dtype=torch.float32)
    _lib.check(L.r2dm_conv2d_ring(x.data_ptr(), w.data_ptr(), b.data_ptr(), packed.data_ptr(), _lib.ptr(aff), prologue,
                                  _lib.ptr(residual), _lib.ptr(sc), y.data_ptr(), B, cin, cout, H, W, k, _st(x)))
    torch.cuda.synchronize()
    return y


def group_norm_affine(x, groups, eps, gamma=None, beta=None, ada=None):
    L = _lib.lib()
    B, C, H, W = x.shape
    x = _lib.f32c(x)
    scratch = torch.empty(L.r2dm_group_norm_scratch_bytes(B, groups), dtype=torch.uint8, device=x.device)
    aff = torch.empty(B, C, 2, device=x.device)
    stats = torch.empty(B, groups, 2, device=x.device)
    _lib.check(L.r2dm_group_norm_affine(x.data_ptr(), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(ada), scratch.data_ptr(),
                                        aff.data_ptr(), stats.data_ptr(), B, C, H, W, groups, eps, _st(x)))
    torch.cuda.synchronize()
    return aff, stats


def affine_act(x, aff, silu):
    L = _lib.lib()
    B, C, H, W = x.shape
    y = torch.empty_like(x)
    _lib.check(L.r2dm_affine_act(x.data_ptr(), aff.data_ptr(), y.data_ptr(), B, C, H * W, int(silu), _st(x)))
    torch.cuda.synchronize()
    return y


def _io(x, io16):
    return x.detach().half().contiguous() if io16 & 1 else _lib.f32c(x)


def fir_down2(x, io16=0):
    """io16 (with R2DM_TEST_IO16 set to the same value by the caller): bit 0 -- x as a half tensor, bit 1 -- y will be one."""
    B, C, H, W = x.shape
    x = _io(x, io16)
    y = torch.empty(B, C, H // 2, W // 2, device=x.device, dtype=torch.float16 if io16 & 2 else torch.float32)
    _lib.check(_lib.lib().r2dm_fir_down2(x.data_ptr(), y.data_ptr(), B, C, H, W, _st(x)))
    torch.cuda.synchronize()
    return y


def fir_down2_stats(x, groups, io16=0):
    """(y, stat): ops.Resample(down=2) plus the GroupNorm statistics of y as the engine's path leaves them -- stat (B, groups, slots, 2)
    float64 [sum, sum of squares]; None where the geometry has no statistics variant."""
    B, C, H, W = x.shape
    slots = _lib.lib().r2dm_fir_down2_stat_slots(C, groups, H, W)
    if slots == 0:
        return None
    x = _io(x, io16)
    y = torch.empty(B, C, H // 2, W // 2, device=x.device, dtype=torch.float16 if io16 & 2 else torch.float32)
    stat = torch.full((B, groups, slots, 2), float("nan"), device=x.device, dtype=torch.float64)
    _lib.check(_lib.lib().r2dm_fir_down2_stats(x.data_ptr(), y.data_ptr(), stat.data_ptr(), B, C, groups, H, W, _st(x)))
    torch.cuda.synchronize()
    return y, stat


def fir_up2(x, io16=0):
    B, C, H, W = x.shape
    x = _io(x, io16)
    y = torch.empty(B, C, H * 2, W * 2, device=x.device, dtype=torch.float16 if io16 & 2 else torch.float32)
    _lib.check(_lib.lib().r2dm_fir_up2(x.data_ptr(), y.data_ptr(), B, C, H, W, _st(x)))
    torch.cuda.synchronize()
    return y


def attention(qkv, heads):
    B, C3, N = qkv.shape
    C = C3 // 3
    out = torch.empty(B, C, N, device=qkv.device)
    _lib.check(_lib.lib().r2dm_attention(qkv.data_ptr(), out.data_ptr(), B, C, heads, N, _st(qkv)))
    torch.cuda.synchronize()
    return out


def time_embedding(cond, freqs, w1, b1, w2, b2):
    B, T, base = cond.shape[0], w1.shape[0], w1.shape[1]
    act = torch.empty(B, T, device=cond.device)
    hid = torch.empty(B, T, device=cond.device)
    _lib.check(_lib.lib().r2dm_time_embedding(cond.data_ptr(), freqs.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                                             w2.data_ptr(), b2.data_ptr(), act.data_ptr(), hid.data_ptr(), B, base, T,
                                             _st(cond)))
    torch.cuda.synchronize()
    return act
