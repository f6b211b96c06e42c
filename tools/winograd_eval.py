#!/usr/bin/env python3
"""Numerical half of the Winograd F(2x2, 3x3) evaluation asked for in VERDICT r2 item 1(iii) (the resource half is DESIGN.md section 10.2).

Emulates, on the CPU in numpy, exactly the arithmetic a gfx950 kernel would do:
  direct   : x, w split into f16 hi/lo, three f16 x f16 products per MAC accumulated in fp32 (what conv_f16s computes)
  winograd : U = G g G^T in fp64 -> fp32 on the host; V = B^T d B in fp32 on the device (additions only); U, V split into f16 hi/lo;
             the 16 element-wise GEMMs with the same 3-term products in fp32; Y = A^T M A in fp32
and prints max / rms error against an fp64 convolution, relative to max|y|, for GroupNorm/GELU-like inputs and He-scaled weights.
No GPU, no cineflow import: python tools/winograd_eval.py [Cin] [Cout] [H]

--row (round 4): the ROW forms instead, as csrc/conv_wino.hip computes them -- the transform runs along x only, ky stays a direct sum:
  F(2,3): V = B^T d per input row (fp32 additions before the split), U = G g per (co, ci, ky) on the host in fp64, four GEMMs of depth 3 Cin,
          y = A^T M in fp32 (the kernel that was built);
  F(4,3): the 6-point form with the interpolation points 0, +-1, +-2, inf (evaluated, not built: 2.25x fewer MFMAs than direct instead of 1.5x)."""
import sys

import numpy as np

G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def split(a32):
    hi = a32.astype(np.float16)
    lo = (a32 - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def mm3(a32, b32):
    """sum_k a[..., m, k] b[..., k, n] with the 3-term f16 split, fp32 accumulation (products of f16 values are exact in fp32)"""
    ah, al = split(a32)
    bh, bl = split(b32)
    return (al @ bh + ah @ bl + ah @ bh).astype(np.float32)


G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)


def row_form(x, w, s, H, g, bt, at):
    """row Winograd F(m, 3) along x: x [cin, H + 2, W + 2 (+ slack)] zero-padded, w [cout, cin, 3, 3]; m = at.shape[0] outputs per unit"""
    m, n = at.shape                                              # outputs per unit, transform points (n = m + 2)
    cout, cin = w.shape[:2]
    T = H // m
    U = np.einsum("jk,ocyk->ocyj", g, w.astype(np.float64) * s).astype(np.float32)              # [o, c, ky, point], host, once
    d = np.stack([x[:, :, m * t:m * t + n] for t in range(T)])                                   # [t, c, row, n]
    V = np.einsum("jk,tcrk->tcrj", bt.astype(np.float32), d).astype(np.float32)                  # fp32, before the split
    M = np.zeros((n, cout, H, T), dtype=np.float32)
    for j in range(n):
        a = U[:, :, :, j].transpose(0, 2, 1).reshape(cout, 3 * cin)                              # k order (ky, channel)
        for y in range(H):
            b = V[:, :, y:y + 3, j].transpose(2, 1, 0).reshape(3 * cin, T)
            M[j, :, y] = mm3(a, b)
    Y = np.einsum("pj,joyt->oytp", at.astype(np.float32), M).astype(np.float32) / s              # [o, y, t, m]
    return Y.reshape(cout, H, T * m), V, M


def main():
    row = "--row" in sys.argv
    if row:
        sys.argv.remove("--row")
    cin = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    cout = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    rng = np.random.default_rng(0)
    x = rng.standard_normal((cin, H + 2, H + 2)).astype(np.float32)
    x = (0.5 * x * (1 + np.tanh(0.79788456 * (x + 0.044715 * x ** 3)))).astype(np.float32)     # GELU-shaped activations
    x[:, 0], x[:, -1], x[:, :, 0], x[:, :, -1] = 0, 0, 0, 0                                       # the zero padding
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    s = 2.0 ** np.floor(np.log2(1024.0 / np.abs(w).max()))                                      # the pack-time power-of-two weight scale
    # ---- fp64 reference
    ref = np.zeros((cout, H, H))
    for ky in range(3):
        for kx in range(3):
            ref += np.einsum("oc,chw->ohw", w[:, :, ky, kx].astype(np.float64), x[:, ky:ky + H, kx:kx + H].astype(np.float64))
    # ---- direct, 3-term split (k order: tap, channel -- the accumulation order of one fp32 accumulator)
    a = (w * s).transpose(0, 2, 3, 1).reshape(cout, 9 * cin)
    cols = np.stack([x[:, ky:ky + H, kx:kx + H] for ky in range(3) for kx in range(3)]).reshape(9 * cin, H * H)
    direct = (mm3(a, cols) / s).reshape(cout, H, H)
    # ---- fp32 direct (what an fp32 convolution gives), for scale
    f32 = (a.astype(np.float32) @ cols.astype(np.float32) / s).reshape(cout, H, H)
    if row:
        scale = np.abs(ref).max()
        rows = [("fp32 direct", f32, None, None), ("3-term split direct (conv_f16s)", direct, None, None)]
        if H % 4 == 0:
            xp = np.pad(x, ((0, 0), (0, 0), (0, 4)))
            rows.append(("3-term split ROW Winograd F(2,3) (conv_wino)",) + row_form(xp, w, s, H, G, BT, AT))
            rows.append(("3-term split ROW Winograd F(4,3) (not built)",) + row_form(xp, w, s, H, G4, BT4, AT4))
        for name, y, V, M in rows:
            e = np.abs(y - ref)
            extra = "" if V is None else "   max|V|/max|x| = %.2f  max|M|/max|y| = %.2f" % (np.abs(V).max() / np.abs(x).max(), np.abs(M).max() / s / scale)
            print("%-46s max|err|/max|y| = %.2e   rms = %.2e%s" % (name, e.max() / scale, np.sqrt((e ** 2).mean()) / scale, extra))
        print("Cin %d Cout %d %dx%d; max|y| = %.3f" % (cin, cout, H, H, scale))
        return
    # ---- Winograd F(2x2, 3x3)
    U = np.einsum("ij,ocjk,lk->ocil", G, w.astype(np.float64) * s, G).astype(np.float32)        # [o, c, 4, 4], host, once
    T = H // 2
    d = np.stack([[x[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4] for tx in range(T)] for ty in range(T)])      # [ty, tx, c, 4, 4]
    bt = BT.astype(np.float32)
    V = np.einsum("ij,yxcjk->yxcik", bt, d).astype(np.float32)                                   # fp32 additions, two stages
    V = np.einsum("yxcik,lk->yxcil", V, bt).astype(np.float32)
    M = np.zeros((4, 4, cout, T * T), dtype=np.float32)
    for i in range(4):
        for j in range(4):
            M[i, j] = mm3(U[:, :, i, j], V[:, :, :, i, j].reshape(T * T, cin).T)
    at = AT.astype(np.float32)
    Y = np.einsum("pi,ijon->pjon", at, M).astype(np.float32)
    Y = np.einsum("pjon,qj->pqon", Y, at).astype(np.float32) / s                                  # [2, 2, o, tiles]
    wino = Y.reshape(2, 2, cout, T, T).transpose(2, 3, 0, 4, 1).reshape(cout, H, H)
    scale = np.abs(ref).max()
    for name, y in (("fp32 direct", f32), ("3-term split direct (conv_f16s)", direct), ("3-term split Winograd F(2x2,3x3)", wino)):
        e = np.abs(y - ref)
        print("%-36s max|err|/max|y| = %.2e   rms = %.2e" % (name, e.max() / scale, np.sqrt((e ** 2).mean()) / scale))
    print("Cin %d Cout %d %dx%d; max|y| = %.3f; max|V| / max|x| = %.2f, max|M| / max|y| = %.2f" %
          (cin, cout, H, H, scale, np.abs(V).max() / np.abs(x).max(), np.abs(M).max() / s / scale))


if __name__ == "__main__":
    main()
