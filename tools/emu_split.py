#!/usr/bin/env python3
"""CPU emulation of the split-precision contractions considered for the conv kernels (evidence for DESIGN.md §2/§4).

For a K-long dot product with operand scales (sa, sw) it reports the error against fp64, normalised by
sum|a*w|, of
  f32      : an fmaf chain in k order (what v_mfma_f32_32x32x2_f32 computes, MI355X guide §3)
  f16x3    : the scheme of csrc/conv_kernel_h3.h (round 2): per-out-channel power-of-two weight scale, hi = RTZ fp16,
             lo = (a - hi) * 2^11 in fp16, third term multiplied against wh * 2^-11; 16-k blocks summed exactly and
             added to an fp32 accumulator
  f16x3_r1 : round 1's scheme (clamp to 65504, unscaled lo) for comparison
  f16, bf16, bf16x2, bf16x3 : single-pass fp16 / bf16 and the 2- and 3-term bf16 splits (3 and 6 products)
Usage: python tools/emu_split.py [--k 2304] [--m 64] [--n 32]
"""
import argparse

import numpy as np


def f16(x):
    return x.astype(np.float16).astype(np.float64)


def f16_rtz(x):
    """float32 -> fp16 round-toward-zero (v_cvt_pkrtz_f16_f32), returned as float64."""
    x = x.astype(np.float32)
    h = x.astype(np.float16)                                   # RTN
    hf = h.astype(np.float32)
    over = np.abs(hf) > np.abs(x)                              # rounded away from zero: step one ulp back
    hb = h.view(np.uint16)
    hb = np.where(over, hb - 1, hb).astype(np.uint16)
    out = hb.view(np.float16).astype(np.float64)
    out = np.where(np.isinf(h) & np.isfinite(x), np.sign(x) * 65504.0, out)
    return out


def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def blocked_acc(terms, block=16):
    """sum over k of the given exact product arrays [M, K, N]-free form: terms is a list of (A [M,K], W [K,N]);
    every 16-k block of every term is summed exactly (fp64) and added to an fp32 accumulator in order."""
    m, k = terms[0][0].shape
    n = terms[0][1].shape[1]
    acc = np.zeros((m, n), np.float32)
    for k0 in range(0, k, block):
        for a, w in terms:
            acc = (acc.astype(np.float64) + a[:, k0:k0 + block] @ w[k0:k0 + block]).astype(np.float32)
    return acc.astype(np.float64)


def fma_chain(a, w):
    acc = np.zeros((a.shape[0], w.shape[1]), np.float32)
    a64, w64 = a.astype(np.float64), w.astype(np.float64)
    for k in range(a.shape[1]):
        acc = (acc.astype(np.float64) + a64[:, k:k + 1] * w64[k:k + 1]).astype(np.float32)
    return acc.astype(np.float64)


def split_weights(w):
    """[K, N] f32 -> (wh, wl, wh_s, inv_scale[N]) as float64 arrays (pack.split_f16x3)."""
    rowmax = np.abs(w).max(axis=0)
    e = np.where(rowmax > 0, np.floor(np.log2(np.maximum(rowmax, 1e-300))), 0.0)
    s = np.where(rowmax > 0, 2.0 ** (13 - e), 1.0)
    ws = w.astype(np.float64) * s
    wh = f16(ws)
    wl = f16(ws - wh)
    wh_s = f16(wh * 2.0 ** -11)
    return wh, wl, wh_s, 1.0 / s


def f16x3(a, w):
    wh, wl, wh_s, inv = split_weights(w)
    ah = f16_rtz(a)
    al = f16((a.astype(np.float64) - ah) * 2048.0)
    return blocked_acc([(ah, wh), (ah, wl), (al, wh_s)]) * inv


def f16x3_r1(a, w):
    ac = np.clip(a, -65504, 65504).astype(np.float32)
    ah = (ac.view(np.uint32) & np.uint32(0xFFFFE000)).view(np.float32).astype(np.float64)
    al = f16_rtz((ac.astype(np.float64) - ah).astype(np.float32))
    ah = f16_rtz(ah.astype(np.float32))
    wc = np.clip(w, -65504, 65504)
    wh = f16(wc)
    wl = f16(wc.astype(np.float64) - wh)
    return blocked_acc([(ah, wh), (ah, wl), (al, wh)])


def bf16_split(x, terms):
    parts, r = [], x.astype(np.float64)
    for _ in range(terms):
        p = bf16(r)
        parts.append(p)
        r = r - p
    return parts


def bf16xn(a, w, n):
    ap, wp = bf16_split(a, n), bf16_split(w, n)
    terms = [(ap[i], wp[j]) for i in range(n) for j in range(n) if i + j < n]
    return blocked_acc(terms)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=2304)
    ap.add_argument("--m", type=int, default=64)
    ap.add_argument("--n", type=int, default=32)
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    print(f"K={args.k}; error = max |x - fp64| / sum|a w| over {args.m}x{args.n} outputs")
    print(f"{'a scale':>8} {'w scale':>8} | {'f32':>9} {'f16x3':>9} {'ratio':>6} | {'f16x3_r1':>9} {'f16':>9} {'bf16':>9} {'bf16x2':>9} {'bf16x3':>9}")
    for sa in (1e-4, 1e-3, 1e-2, 1.0, 1e2, 1e4):
        for sw in (2e-4, 2e-3, 2e-2, 1.0):
            a = (rng.standard_normal((args.m, args.k)) * sa).astype(np.float32)
            w = (rng.standard_normal((args.k, args.n)) * sw).astype(np.float32)
            exact = a.astype(np.float64) @ w.astype(np.float64)
            den = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64)
            err = lambda x: float((np.abs(x - exact) / den).max())
            e32, e3 = err(fma_chain(a, w)), err(f16x3(a, w))
            print(f"{sa:8.0e} {sw:8.0e} | {e32:9.2e} {e3:9.2e} {e3 / e32:6.2f} | {err(f16x3_r1(a, w)):9.2e} "
                  f"{err(blocked_acc([(f16(a), f16(w))])):9.2e} {err(blocked_acc([(bf16(a), bf16(w))])):9.2e} "
                  f"{err(bf16xn(a, w, 2)):9.2e} {err(bf16xn(a, w, 3)):9.2e}")


if __name__ == "__main__":
    main()
