#!/usr/bin/env python3
"""CPU numerics probe (numpy): would a Winograd F(2x2, 3x3) formulation of the 3x3 convolution on the fp16 matrix pipe keep
fp32-class accuracy?  Direct vs Winograd, both with the f16x2 operand split (fp16 piece + 2^11-scaled fp16 residual, h*h
products in one fp32 accumulator, cross products in a second one, one rounding per 16-k MFMA step), against fp64.
Input: silu(N(0,1)) activations, N(0, 1/(9 Cin)) weights -- the residual blocks' operating point."""
import numpy as np
rng = np.random.default_rng(0)
def silu(x): return x / (1 + np.exp(-x))
def split(v, S=2048.0):
    h = v.astype(np.float16).astype(np.float32)
    l = ((v.astype(np.float32) - h) * S).astype(np.float16).astype(np.float32)
    return h, l
def mfma_gemm(A, B):  # A (M,K), B (K,N) fp32 values already representable; returns hi + lo/2048 with per-16-k fp32 rounding
    Ah, Al = split(A); Bh, Bl = split(B)
    hi = np.zeros((A.shape[0], B.shape[1]), np.float32); lo = hi.copy()
    for k in range(0, A.shape[1], 16):
        s = slice(k, k + 16)
        lo = (lo.astype(np.float64) + Al[:, s].astype(np.float64) @ Bh[s].astype(np.float64)).astype(np.float32)
        lo = (lo.astype(np.float64) + Ah[:, s].astype(np.float64) @ Bl[s].astype(np.float64)).astype(np.float32)
        hi = (hi.astype(np.float64) + Ah[:, s].astype(np.float64) @ Bh[s].astype(np.float64)).astype(np.float32)
    return (hi + lo / np.float32(2048)).astype(np.float32)
Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float32)
At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)
for Cin in (64, 256):
    Cout, H, W = 32, 8, 16
    x = silu(rng.standard_normal((Cin, H + 2, W + 2))).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    # fp64 truth
    truth = np.zeros((Cout, H, W))
    for ky in range(3):
        for kx in range(3):
            truth += np.einsum('oc,chw->ohw', w[:, :, ky, kx].astype(np.float64), x[:, ky:ky + H, kx:kx + W].astype(np.float64))
    sc = np.sqrt((truth ** 2).mean())
    # fp32 direct (einsum in float32)
    f32 = np.zeros((Cout, H, W), np.float32)
    for ky in range(3):
        for kx in range(3):
            f32 += np.einsum('oc,chw->ohw', w[:, :, ky, kx], x[:, ky:ky + H, kx:kx + W]).astype(np.float32)
    # direct f16x2: implicit GEMM, K = (tap, channel) in chunks of 16 channels per tap
    A = w.transpose(0, 2, 3, 1).reshape(Cout, 9 * Cin)                       # (co, tap*Cin)
    cols = np.stack([x[:, ky:ky + H, kx:kx + W].reshape(Cin, H * W) for ky in range(3) for kx in range(3)]).reshape(9 * Cin, H * W)
    direct = mfma_gemm(A, cols).reshape(Cout, H, W)
    # Winograd F(2x2,3x3): U = G g G^T (fp32, at load), V = B^T d B (fp32), 16 GEMMs over channels, Y = A^T M A (fp32)
    U = np.einsum('ij,ocjk,lk->ocil', G, w, G).astype(np.float32)                   # (co, ci, 4, 4)
    th, tw = H // 2, W // 2
    d = np.stack([[x[:, 2 * i:2 * i + 4, 2 * j:2 * j + 4] for j in range(tw)] for i in range(th)])  # (th, tw, ci, 4, 4)
    V = np.einsum('ij,abcjk,lk->abcil', Bt, d, Bt).astype(np.float32)              # (th, tw, ci, 4, 4)
    M = np.zeros((Cout, th * tw, 4, 4), np.float32)
    for a in range(4):
        for b in range(4):
            M[:, :, a, b] = mfma_gemm(U[:, :, a, b], V[:, :, :, a, b].reshape(th * tw, Cin).T)
    Y = np.einsum('ij,otjk,lk->otil', At, M, At).astype(np.float32)               # (co, tiles, 2, 2)
    wino = Y.reshape(Cout, th, tw, 2, 2).transpose(0, 1, 3, 2, 4).reshape(Cout, H, W)
    e = lambda y: np.sqrt(((y - truth) ** 2).mean()) / sc
    print(f"Cin={Cin:4d}: rel rms error vs fp64   fp32 einsum {e(f32):.2e}   direct f16x2 {e(direct):.2e}   Winograd F(2x2,3x3) f16x2 {e(wino):.2e}   (max |wino-truth| {np.abs(wino - truth).max() / sc:.2e})")
