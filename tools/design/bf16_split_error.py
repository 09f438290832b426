#!/usr/bin/env python3
"""Design study (CPU, numpy): how exact is a float32 product sum evaluated as bf16 x bf16 MFMA terms with float32 accumulation?

DESIGN section 8 names "exact f32 products out of split bf16 operands" as the one lever left on the update kernels (the bf16 MFMA
rate is 16 x the f32 rate).  Before anyone builds it, this measures what it would do to the numbers, on the shapes the update
kernel multiplies (K = 16 and K = 64 contractions over hidden units, K = 32 over the samples of a tile), against float64:

  f32        : the product sum as the f32 MFMA evaluates it (float32 products and running sum, k in order)
  split-n/t  : every operand split into n bf16 pieces (a = a0 + a1 + ..., each the bf16 rounding of what is left); the t lowest-order
               piece products are dropped; each kept term is an exact bf16 x bf16 product summed in float32 in the MFMA's k order.
               3/6 = six terms (a0b0, a0b1, a1b0, a0b2, a1b1, a2b0): 6 bf16 MFMAs at 16 x the rate = 2.7 x the f32 MFMA throughput.

Only numpy; bf16 rounding is round-to-nearest-even on the upper 16 bits of the float32 pattern, as v_cvt_pk_bf16_f32 does.
usage: python tools/design/bf16_split_error.py"""
import numpy as np


def bf16(x):
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def split(x, n):
    parts, rest = [], np.asarray(x, dtype=np.float32).copy()
    for _ in range(n):
        p = bf16(rest)
        parts.append(p)
        rest = (rest - p).astype(np.float32)   # exact: p is the leading bits of rest
    return parts


def dot_f32(a, b):
    """sum_k a[m, k] b[k, n] with float32 products and a float32 running sum, k in order (what a chain of f32 MFMAs does)."""
    acc = np.zeros((a.shape[0], b.shape[1]), dtype=np.float32)
    for k in range(a.shape[1]):
        acc = (acc + (a[:, k:k + 1] * b[k:k + 1, :]).astype(np.float32)).astype(np.float32)
    return acc


def dot_split(a, b, n, keep):
    """keep = list of (i, j) piece pairs; term order = MFMA issue order, each term's k in order, all into one f32 accumulator."""
    pa, pb = split(a, n), split(b, n)
    acc = np.zeros((a.shape[0], b.shape[1]), dtype=np.float32)
    for i, j in keep:
        for k in range(a.shape[1]):
            prod = pa[i][:, k:k + 1].astype(np.float64) * pb[j][k:k + 1, :].astype(np.float64)   # bf16 x bf16 is exact in f32
            acc = (acc + prod.astype(np.float32)).astype(np.float32)
    return acc


def report(name, a, b):
    ref = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(a.astype(np.float64)) @ np.abs(b.astype(np.float64))   # sum_k |a||b|: the natural error scale of a dot product
    rows = [("f32", dot_f32(a, b))]
    low_first = [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)]           # small terms first: they are not swallowed by the big one
    rows.append(("split-2/3 terms", dot_split(a, b, 2, [(1, 0), (0, 1), (0, 0)])))
    rows.append(("split-3/6 terms", dot_split(a, b, 3, low_first)))
    rows.append(("split-3/6, big first", dot_split(a, b, 3, low_first[::-1])))
    rows.append(("split-3/9 terms", dot_split(a, b, 3, [(2, 2), (2, 1), (1, 2)] + low_first)))
    print(f"{name}: A {a.shape} x B {b.shape}")
    for label, got in rows:
        err = np.abs(got.astype(np.float64) - ref)
        print(f"  {label:22s} max |err| / sum|a||b| = {np.max(err / scale):.3e}   rms = {np.sqrt(np.mean((err / scale) ** 2)):.3e}"
              f"   max |err| / |ref| (|ref| > 1e-3 scale) = {np.max(np.where(np.abs(ref) > 1e-3 * scale, err / np.abs(ref), 0)):.3e}")


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    # forward: weights ~ U(-1/sqrt(K), 1/sqrt(K)) (nn.Linear init), activations relu-like (half zeros), observations in [0, 1]
    for K, what in ((16, "layer 1 (K = 16 observation entries)"), (64, "layer 2 (K = 64 hidden units)")):
        w = rng.uniform(-1, 1, (64, K)).astype(np.float32) / np.float32(np.sqrt(K))
        x = np.maximum(rng.normal(0.3, 0.5, (K, 512)), 0).astype(np.float32)
        report(what, w, x)
    # weight gradient: contraction over the 32 samples of a tile, one operand a gradient with a wide dynamic range
    g = (rng.normal(0, 1, (64, 32)) * 10.0 ** rng.uniform(-6, -2, (64, 32))).astype(np.float32)
    h = np.maximum(rng.normal(0.3, 0.5, (32, 64)), 0).astype(np.float32)
    report("weight gradient (K = 32 samples, gradients over four decades)", g, h)
    print("float32 unit roundoff 2^-24 = 5.96e-08; bf16 piece: 8 bits, three pieces = 24 bits")
