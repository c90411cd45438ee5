#!/usr/bin/env python
"""Numerics of the plan in profiles/r1_notes.md ("BatchNorm backward of the wide unit by linearity").

For the last unit of a bottleneck block (y3 = z2 W3^T, 1x1 conv; BatchNorm over y3) the backward
tensor dy3 = ca*g + cb*y3 + cc never has to be materialised:

    dz2 = g (diag(ca) W3) + z2 Q + 1 (cc^T W3),          Q = W3^T diag(cb) W3
    dW3 = diag(ca) (g^T z2) + diag(cb) W3 (z2^T z2) + cc colsum(z2)^T
    sum_m g*y3 = rowsum(W3 o (g^T z2))

This script compares both formulations, with every GEMM operand rounded to bf16 where the kernels
would round it, against an fp64 evaluation of the exact BatchNorm backward.  CPU only.
"""
import argparse

import torch


def bf(x):
    return x.to(torch.bfloat16).to(torch.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=32768)
    ap.add_argument("--mid", type=int, default=64)
    ap.add_argument("--seed", type=int, default=0)
    ns = ap.parse_args()
    torch.manual_seed(ns.seed)
    M, C, K = ns.rows, ns.mid, 4 * ns.mid
    z2 = bf(torch.relu(torch.randn(M, C, dtype=torch.float64)))
    W3 = bf(torch.randn(K, C, dtype=torch.float64) / C ** 0.5)
    y3_exact = z2 @ W3.t()
    y3 = bf(y3_exact)                       # what the forward conv stored
    g = bf(torch.randn(M, K, dtype=torch.float64) * 1e-3 * (1 + torch.rand(K, dtype=torch.float64)))
    gamma = 0.5 + torch.rand(K, dtype=torch.float64)
    eps = 1e-5

    def coeffs(y, sgy):
        mean = y.mean(0)
        var = y.var(0, unbiased=False)
        inv = (var + eps).rsqrt()
        s1 = g.sum(0)
        dgamma = inv * (sgy - mean * s1)
        ca = gamma * inv
        cb = -ca * inv * dgamma / M
        cc = -ca * s1 / M - cb * mean
        return ca, cb, cc

    # ground truth: exact arithmetic on the stored tensors
    ca, cb, cc = coeffs(y3, (g * y3).sum(0))
    dy3_true = ca * g + cb * y3 + cc
    dz2_true = dy3_true @ W3
    dW3_true = dy3_true.t() @ z2

    def err(a, ref):
        return float(((a - ref).pow(2).mean() / ref.pow(2).mean()).sqrt())

    # today's kernels: dy3 rounded to bf16, then two bf16 GEMMs with fp32 accumulation
    dy3_std = bf(dy3_true)
    print("standard   dz2 %.2e   dW3 %.2e" % (err(dy3_std @ W3, dz2_true), err(dy3_std.t() @ z2, dW3_true)))

    # linearity: raw weight gradient first, everything else from it
    raw = g.t() @ z2                                    # wgrad on the un-normalised gradient (fp32 slabs)
    sgy = (W3 * raw).sum(1)                             # sum_m g*y3 without reading y3
    ca2, cb2, cc2 = coeffs(y3, sgy)                     # mean / var still come from the forward statistics
    print("sum g*y3   %.2e   (relative to the direct sum)" % err(sgy, (g * y3).sum(0)))
    A = bf(ca2[:, None] * W3)                           # prepared weights, bf16
    Q = bf(W3.t() @ (cb2[:, None] * W3))
    dz2_lin = g @ A + z2 @ Q + (cc2 @ W3)[None, :]
    gram = z2.t() @ z2                                  # fp32 slabs as well
    dW3_lin = ca2[:, None] * raw + cb2[:, None] * (W3 @ gram) + cc2[:, None] * z2.sum(0)[None, :]
    print("linearity  dz2 %.2e   dW3 %.2e" % (err(dz2_lin, dz2_true), err(dW3_lin, dW3_true)))
    Qb = bf(gram)
    dW3_lin_b = ca2[:, None] * raw + cb2[:, None] * (W3 @ Qb) + cc2[:, None] * z2.sum(0)[None, :]
    print("linearity with a bf16 Gram matrix: dW3 %.2e" % err(dW3_lin_b, dW3_true))


if __name__ == "__main__":
    main()
