"""GPU parity tests of every C-ABI operator against the oracle (and the golden vectors produced by the reference).

Tolerances: the warp family reproduces the reference's fp32 rounding path, so 2e-6 abs on O(1) data; GEMM-shaped
ops differ from the oracle only by fp32 summation order (the MFMA is an exact fp32 fmaf chain), so 1e-5 relative to
the operand magnitude; label maps must be identical."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def maxdiff(a, b):
    return float((a.detach().cpu().double() - torch.as_tensor(b).double()).abs().max())


def check(a, b, tol, what=""):
    d = maxdiff(a, b)
    assert d <= tol, "%s max|diff| %.3e > %.1e" % (what, d, tol)


def randn(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


# ------------------------------------------------------------------------------------------------ warp family
@pytest.mark.parametrize("tag", ["32", "40x24", "3d"])
def test_warp_golden(dev, golden, tag):
    from cineflow import ops
    g = golden("warp_" + tag)
    flow, src = T(g["flow"]).to(dev), T(g["src"]).to(dev)
    check(ops.warp_bilinear(flow, src), g["warped"], 2e-6, "warp")
    check(ops.vecint(flow, 7), g["vecint"], 1e-5, "vecint")


def test_warp_256_golden_and_properties(dev, golden):
    from cineflow import ops
    from oracle import ops as OO
    g = golden("warp_256")
    src = randn(1, 4, 256, 256, seed=13)
    flow = F.avg_pool2d(randn(1, 2, 288, 288, seed=12), 33, stride=1) * 120.0
    out = ops.warp_bilinear(flow.to(dev), src.to(dev))
    check(out[:, :, :16, :16], g["warped_corner"], 2e-6)
    check(out[:, :, 120:136, 120:136], g["warped_center"], 2e-6)
    assert abs(float(out.double().sum()) - float(g["checksum"])) < 1e-2
    check(out, OO.warp_bilinear(flow, src), 2e-6)
    # identity: zero flow returns the source up to the reference's own normalise round trip (3.8e-5 at 256, SURVEY B)
    ident = ops.warp_bilinear(torch.zeros(1, 2, 256, 256, device=dev), src.to(dev))
    check(ident, OO.warp_bilinear(torch.zeros(1, 2, 256, 256), src), 2e-6)
    check(ident, src, 1e-4)
    # linearity in the source
    a, b = randn(1, 2, 256, 256, seed=1).to(dev), randn(1, 2, 256, 256, seed=2).to(dev)
    f = flow.to(dev)
    lhs = ops.warp_bilinear(f, ops.add(a, b))
    rhs = ops.add(ops.warp_bilinear(f, a), ops.warp_bilinear(f, b))
    check(lhs, rhs.cpu(), 1e-5)


def test_warp_edge_cases(dev):
    from cineflow import ops
    from oracle import ops as OO
    # flows that leave the image on every side, exactly-integer displacements, huge and NaN-free extremes
    H, W = 17, 23
    src = randn(2, 3, H, W, seed=5)
    flow = torch.zeros(2, 2, H, W)
    flow[0, 0] = 5.0
    flow[0, 1] = -7.0
    flow[1] = 40.0 * randn(2, H, W, seed=6)
    flow[1, :, 0, 0] = 1e6
    check(ops.warp_bilinear(flow.to(dev), src.to(dev)), OO.warp_bilinear(flow.clone(), src), 2e-5)


def test_warp_labels(dev, golden):
    from cineflow import ops
    from oracle import ops as OO
    g = golden("warp_labels")
    flow = T(g["flow"])
    labels = T(g["labels"])  # [B,1,H,W] float
    out = ops.warp_labels(flow.to(dev), labels[:, 0].to(torch.uint8).to(dev))
    ref = T(g["registered"])[:, :, 0].to(torch.uint8)
    agree = float((out.cpu() == ref).float().mean())
    assert agree == 1.0, agree
    # 256x256, 4 concentric rings, smooth flow (BASELINE config 2)
    yy, xx = np.mgrid[:256, :256]
    rad = np.sqrt((yy - 127.5) ** 2 + (xx - 127.5) ** 2)
    lab = np.zeros((256, 256), np.uint8)
    for k, r in enumerate((100, 70, 40), start=1):
        lab[rad < r] = k
    labels = torch.from_numpy(lab)[None]
    flow = (F.avg_pool2d(randn(1, 2, 288, 288, seed=21), 33, stride=1) * 200.0)[None]
    out = ops.warp_labels(flow.to(dev), labels.to(dev)).cpu()
    ref = OO.warp_labels(flow, labels[:, None].float())[:, :, 0].to(torch.uint8)
    mism = int((out != ref).sum())
    assert mism <= 2, "label mismatches %d (ties at class borders only)" % mism
    for k in (1, 2, 3):
        assert abs(OO.dice(out.numpy(), ref.numpy(), k) - 1.0) <= 1e-3


def test_warp_family_four_pixel_kernels_full_size(dev):
    """The W % 4 == 0 kernels (four pixels per thread, taps through a buffer resource) at 256x256 and at a width that is not a multiple of
    four (one-pixel kernels): the Jacobian is bit-identical to numpy (every product and sum rounds on its own: warp.hip is built with
    -ffp-contract=off), the label warp agrees with the oracle on every pixel of a random label map, the image warp is within 2 ulp of ATen."""
    from cineflow import ops
    from oracle import ops as OO
    for (B, C, H, W, amp) in [(3, 2, 256, 256, 3.0), (2, 1, 40, 24, 5.0), (2, 2, 33, 47, 8.0)]:
        flow, src = amp * randn(B, 2, H, W, seed=H), randn(B, C, H, W, seed=W)
        out = ops.warp_bilinear(flow.to(dev), src.to(dev)).cpu()
        check(out, OO.warp_bilinear(flow.clone(), src), 1e-6, "warp %dx%d" % (H, W))
        det = ops.jacobian_det(flow.to(dev)).cpu().numpy()
        ref = np.stack([OO.jacobian_determinant(flow[b].permute(1, 2, 0).numpy().astype(np.float64)) for b in range(B)])
        assert det.dtype == np.float64 and np.array_equal(det, ref), "jacobian %dx%d: max|diff| %.3e" % (H, W, np.abs(det - ref).max())
        lab = (torch.rand(B, H, W, generator=torch.Generator().manual_seed(3)) * 4).to(torch.uint8)
        wl = ops.warp_labels(flow[None].to(dev), lab.to(dev)).cpu()
        wr = OO.warp_labels(flow[None], lab[:, None].float())[:, :, 0]
        assert int((wl.long() != wr).sum()) == 0, "label warp %dx%d" % (H, W)
    # taps outside the image on every side (zeros), through the out-of-range buffer offsets
    flow = torch.zeros(1, 2, 16, 32)
    flow[0, 0, :4], flow[0, 0, -4:], flow[0, 1, :, :8], flow[0, 1, :, -8:] = -9.5, 9.5, -11.25, 40.75
    src = randn(1, 3, 16, 32, seed=77)
    check(ops.warp_bilinear(flow.to(dev), src.to(dev)), OO.warp_bilinear(flow.clone(), src), 1e-6, "borders")
    lab = (torch.rand(1, 16, 32, generator=torch.Generator().manual_seed(4)) * 4).to(torch.uint8)
    assert torch.equal(ops.warp_labels(flow[None].to(dev), lab.to(dev)).cpu().long(), OO.warp_labels(flow[None], lab[:, None].float())[:, :, 0])


def test_warp_3d_properties(dev):
    """3-D branch at a realistic volume size: zero flow is the identity, a pure integer shift moves the volume and zero-fills,
    and the oracle agrees on a random field."""
    from cineflow import ops
    from oracle import ops as OO
    B, C, D, H, W = 1, 2, 12, 96, 80
    src = randn(B, C, D, H, W, seed=40)
    out = ops.warp_bilinear(torch.zeros(B, 3, D, H, W, device=dev), src.to(dev)).cpu()
    check(out, src, 1e-4, "identity (the reference's normalise round trip is not exact either: 3.8e-5 at 256, SURVEY 8a a16)")
    flow = torch.zeros(B, 3, D, H, W)
    flow[:, 0], flow[:, 1], flow[:, 2] = 2.0, -3.0, 5.0
    out = ops.warp_bilinear(flow.to(dev), src.to(dev)).cpu()
    ref = torch.zeros_like(src)
    ref[:, :, :D - 2, 3:, :W - 5] = src[:, :, 2:, :H - 3, 5:]
    check(out, ref, 2e-4, "integer shift")
    flow = 2.0 * randn(B, 3, D, H, W, seed=41)
    check(ops.warp_bilinear(flow.to(dev), src.to(dev)), OO.warp_bilinear(flow.clone(), src), 5e-5, "random field (coordinate rounding scales with the extent: eps*96 px)")


def test_memory_input(dev):
    from cineflow import ops
    from oracle import ops as OO
    B, H, W = 3, 40, 56
    x0, xt, cum = randn(B, 1, H, W, seed=7), randn(B, 1, H, W, seed=8), 3 * randn(B, 2, H, W, seed=9)
    out = ops.memory_input(x0.to(dev), xt.to(dev), cum.to(dev))
    reg = OO.warp_bilinear(cum.clone(), xt)
    check(out, torch.cat([x0, xt, cum, x0 - reg, reg], 1), 2e-6)


def test_jacobian(dev, golden):
    from cineflow import ops
    from oracle import ops as OO
    g = golden("jacobian")
    disp = T(g["disp"]).float()  # [H,W,2]
    det = ops.jacobian_det(disp.permute(2, 0, 1)[None].contiguous().to(dev))
    check(det[0], OO.jacobian_determinant(disp.numpy().astype(np.float64)), 1e-12)
    check(ops.jacobian_det(torch.zeros(2, 2, 9, 5, device=dev)), np.ones((2, 9, 5)), 0)
    big = 4 * randn(2, 2, 256, 256, seed=10)
    ref = np.stack([OO.jacobian_determinant(big[b].permute(1, 2, 0).numpy().astype(np.float64)) for b in range(2)])
    check(ops.jacobian_det(big.to(dev)), ref, 1e-10)
    # 3-D case (compute_jacobian.py:42-52)
    disp3 = T(g["disp3"]).float()  # [D,H,W,3]
    det3 = ops.jacobian_det(disp3.permute(3, 0, 1, 2)[None].contiguous().to(dev))
    check(det3[0], OO.jacobian_determinant(disp3.numpy().astype(np.float64)), 1e-12)
    check(ops.jacobian_det(torch.zeros(1, 3, 4, 9, 5, device=dev)), np.ones((1, 4, 9, 5)), 0)
    big3 = 2 * randn(1, 3, 10, 64, 48, seed=42)
    ref3 = OO.jacobian_determinant(big3[0].permute(1, 2, 3, 0).numpy().astype(np.float64))
    check(ops.jacobian_det(big3.to(dev))[0], ref3, 1e-10)


# ------------------------------------------------------------------------------------------------ correlation
@pytest.mark.parametrize("C,H,W,stride", [(16, 32, 64, 1), (24, 40, 72, 2), (8, 48, 80, 4), (64, 64, 64, 1), (12, 19, 37, 4),
                                          (5, 7, 9, 2), (12, 20, 40, 4), (20, 30, 72, 1), (36, 50, 132, 2), (8, 256, 256, 4),
                                          (3, 130, 200, 1)])
def test_corr_volume_radius4(dev, C, H, W, stride):
    from cineflow import ops
    from oracle import ops as OO
    cur, prev = randn(2, C, H, W, seed=11), randn(2, C, H, W, seed=12)
    check(ops.corr_volume(cur.to(dev), prev.to(dev), 4, stride), OO.corr_volume(cur, prev, 4, stride), 1e-5, "corr_volume")


@pytest.mark.parametrize("B,C,H,W,stride", [(2, 64, 256, 256, 4), (3, 128, 128, 128, 2), (2, 256, 64, 64, 1),      # the three levels of the flow network
                                            (5, 16, 64, 128, 4), (3, 32, 48, 192, 2), (9, 16, 24, 64, 1),             # ragged tile counts, bands with < 1 tile
                                            (9, 64, 256, 256, 4), (20, 128, 128, 128, 2), (70, 256, 64, 64, 1)])      # 4-5 tiles per workgroup: the chunk stream runs across tile (and sample) boundaries
def test_corr_volume_mfma_vs_oracle_and_vector_kernel(dev, B, C, H, W, stride):
    """the f16-MFMA CorrVolume kernel (csrc/corr_mfma.hip: 2-D banded products of hi/lo-split operands, four split terms inside
    one K = 32 instruction, fp32 accumulation) at the network's three levels with B >= 2, against the oracle at 1e-5 and against the fp32
    vector kernel (cf_corr_mfma_enable(0)) on the same inputs; zero padding at every border (dilation 4: +-16 px)."""
    from cineflow import ops
    from cineflow._lib import lib
    from oracle import ops as OO
    cur, prev = randn(B, C, H, W, seed=21), randn(B, C, H, W, seed=22)
    prev_mode = lib().cf_corr_mfma_enable(1)
    try:
        got = ops.corr_volume(cur.to(dev), prev.to(dev), 4, stride)
        lib().cf_corr_mfma_enable(0)
        vec = ops.corr_volume(cur.to(dev), prev.to(dev), 4, stride)
    finally:
        lib().cf_corr_mfma_enable(prev_mode)
    want = OO.corr_volume(cur[:2], prev[:2], 4, stride)
    check(got[:2], want, 1e-5, "corr_volume (MFMA) vs oracle")
    check(got, vec.cpu(), 1e-5, "corr_volume (MFMA) vs the fp32 vector kernel")
    assert got.shape == (B, 81, H, W) and bool(torch.isfinite(got).all())


def test_corr_volume_generic_and_symmetry(dev):
    from cineflow import ops
    from oracle import ops as OO
    cur, prev = randn(1, 6, 20, 24, seed=13), randn(1, 6, 20, 24, seed=14)
    check(ops.corr_volume(cur.to(dev), prev.to(dev), 2, 3), OO.corr_volume(cur, prev, 2, 3), 1e-5)
    # size-independent property at a BASELINE level: corr(cur,prev)[d](p) == corr(prev,cur)[-d](p + d*s)
    C, S, s = 32, 128, 2
    a, b = randn(1, C, S, S, seed=15).to(dev), randn(1, C, S, S, seed=16).to(dev)
    ab, ba = ops.corr_volume(a, b, 4, s).cpu(), ops.corr_volume(b, a, 4, s).cpu()
    for (dy, dx) in [(1, 2), (-4, 4), (3, -1)]:
        ch, chm = (dy + 4) * 9 + (dx + 4), (-dy + 4) * 9 + (-dx + 4)
        y0, y1 = max(0, -dy * s), min(S, S - dy * s)
        x0, x1 = max(0, -dx * s), min(S, S - dx * s)
        lhs = ab[0, ch, y0:y1, x0:x1]
        rhs = ba[0, chm, y0 + dy * s:y1 + dy * s, x0 + dx * s:x1 + dx * s]
        check(lhs, rhs, 1e-5)


def test_allpairs_pyramid_lookup(dev):
    from cineflow import ops
    from oracle import ops as OO
    B, C, H, W = 2, 32, 16, 16
    f1, f2 = randn(B, C, H, W, seed=17), randn(B, C, H, W, seed=18)
    pyr = ops.corr_pyramid(f1.to(dev), f2.to(dev), 3)
    ref = OO.corr_pyramid(OO.corr_allpairs(f1, f2), 3)
    off = 0
    for l, r in enumerate(ref):
        n = r.numel()
        check(pyr[off:off + n].view(r.shape), r, 2e-5, "level %d" % l)
        off += n
    coords = OO.coords_grid(B, H, W) + 2.5 * randn(B, 2, H, W, seed=19)
    out = ops.corr_lookup(pyr, coords.to(dev), 3, 4)
    check(out, OO.corr_lookup(ref, coords, 4), 2e-5, "lookup")
    check(ops.coords_grid(B, H, W, dev), OO.coords_grid(B, H, W), 0)


def test_allpairs_pyramid_fused_full_size(dev):
    """RAFT at 1/8 of 256 x 256: 256-channel 32 x 32 maps, 4 levels -- the fused f16-split kernel (level 0 and the three pooled levels from
    one set of accumulators), then the tiled lookup, against the oracle; and the fused pooled levels against pooling the stored level 0
    (bit-compatible by construction)."""
    import torch.nn.functional as F_
    from cineflow import ops
    from oracle import ops as OO
    B, C, H, W = 3, 256, 32, 32
    f1, f2 = randn(B, C, H, W, seed=27), randn(B, C, H, W, seed=28)
    pyr = ops.corr_pyramid(f1.to(dev), f2.to(dev), 4)
    ref = OO.corr_pyramid(OO.corr_allpairs(f1, f2), 4)
    off, lv = 0, []
    for l, r in enumerate(ref):
        n = r.numel()
        lv.append(pyr[off:off + n].view(r.shape).cpu())
        check(lv[-1], r, 3e-5, "level %d" % l)
        off += n
    for l in range(1, 4):
        assert torch.equal(lv[l], F_.avg_pool2d(lv[l - 1], 2, stride=2)), "level %d is not the pool of the stored level %d" % (l, l - 1)
    coords = OO.coords_grid(B, H, W) + 3.0 * randn(B, 2, H, W, seed=29)
    out = ops.corr_lookup(pyr, coords.to(dev), 4, 4)
    check(out, OO.corr_lookup(ref, coords, 4), 3e-5, "lookup")


def test_convex_upsample(dev):
    from cineflow import ops
    from oracle import ops as OO
    flow, mask = randn(2, 2, 8, 12, seed=20), 2 * randn(2, 576, 8, 12, seed=21)
    check(ops.convex_upsample(flow.to(dev), mask.to(dev)), OO.convex_upsample(flow, mask), 1e-5)
    seg = randn(1, 4, 8, 8, seed=22)
    check(ops.convex_upsample(seg.to(dev), mask[:1, :, :, :8].contiguous().to(dev)), OO.convex_upsample(seg, mask[:1, :, :, :8]), 1e-5)
    # the RAFT map size of the bench (32 x 32 -> 256 x 256; the row kernel: one workgroup per map row) and a 40-wide map (a second, partly
    # filled 32-column block per row); C = 2 takes the row kernel, C = 4 above the flat one
    for (B, h, w, seed) in ((3, 32, 32, 23), (2, 6, 40, 25)):
        flow, mask = randn(B, 2, h, w, seed=seed), 2 * randn(B, 576, h, w, seed=seed + 1)
        check(ops.convex_upsample(flow.to(dev), mask.to(dev)), OO.convex_upsample(flow, mask), 1e-5, "convex upsample %dx%d" % (h, w))


# ------------------------------------------------------------------------------------------------ conv
CONV_CASES = [
    # B, C1, C2, H, W, Cout, kh, kw, stride, pad
    (2, 6, 0, 32, 32, 16, 3, 3, 1, (1, 1)),
    (2, 16, 0, 32, 32, 32, 3, 3, 2, (1, 1)),
    (1, 8, 8, 20, 28, 24, 3, 3, 1, (1, 1)),      # dual input (cat)
    (3, 1, 0, 17, 13, 5, 3, 3, 1, (1, 1)),       # ragged everything, K odd
    (2, 64, 0, 16, 16, 64, 1, 1, 1, (0, 0)),
    (2, 16, 0, 16, 16, 40, 1, 1, 2, (0, 0)),     # strided 1x1 (residual downsample)
    (1, 2, 0, 16, 16, 128, 7, 7, 1, (3, 3)),
    (1, 24, 8, 12, 20, 16, 1, 5, 1, (0, 2)),
    (1, 24, 8, 12, 20, 16, 5, 1, 1, (2, 0)),
    (1, 96, 0, 8, 8, 160, 3, 3, 1, (1, 1)),      # Cout > 128, small map
    (1, 33, 0, 4, 4, 70, 3, 3, 1, (1, 1)),       # 4x4 map (Generic_UNet bottleneck)
    (2, 64, 0, 64, 64, 2, 3, 3, 1, (1, 1)),      # flow head
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(dev, case):
    from cineflow import ops
    B, C1, C2, H, W, Cout, kh, kw, stride, pad = case
    x1 = randn(B, C1, H, W, seed=30)
    x2 = randn(B, C2, H, W, seed=31) if C2 else None
    w = randn(Cout, C1 + C2, kh, kw, seed=32) / math.sqrt((C1 + C2) * kh * kw)
    b = randn(Cout, seed=33)
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    ref = F.conv2d(xin, w, b, stride=stride, padding=pad)
    out = ops.conv2d(x1.to(dev), ops.prep_conv_weight(w).to(dev), b.to(dev), Cout, kh, kw, stride, pad, x2=None if x2 is None else x2.to(dev))
    check(out, ref, 2e-5, "conv")
    # fused epilogue: activation + residual + channel offset
    res = randn(*ref.shape, seed=34)
    big = torch.full((B, Cout + 3) + tuple(ref.shape[2:]), 7.0, device=dev)
    ops.conv2d(x1.to(dev), ops.prep_conv_weight(w).to(dev), b.to(dev), Cout, kh, kw, stride, pad, x2=None if x2 is None else x2.to(dev),
               act="gelu", res=res.to(dev), out=big, out_coff=2)
    check(big[:, 2:2 + Cout], F.gelu(ref) + res, 2e-5, "conv epilogue")
    assert float(big[:, :2].min()) == 7.0 and float(big[:, 2 + Cout:].max()) == 7.0


F16S_CASES = [c for c in CONV_CASES if (c[6], c[7]) in ((3, 3), (1, 1), (1, 5), (5, 1))] + [
    (2, 128, 256, 32, 32, 256, 1, 5, 1, (0, 2)), # SepConvGRU [r | z] gates, horizontal pass, full width (8-wave shape, vector staging)
    (2, 128, 256, 32, 32, 128, 5, 1, 1, (2, 0)), # SepConvGRU candidate, vertical pass
    (3, 40, 24, 16, 16, 48, 1, 5, 1, (0, 2)),    # 64-channel shape, split not a chunk multiple (split-aware packing), ragged batch
    (3, 40, 24, 8, 8, 48, 5, 1, 1, (2, 0)),      # 8x8 maps: several images per workgroup
    (1, 32, 0, 13, 18, 20, 5, 1, 1, (2, 0)),     # W % 4 != 0: scalar staging
    (1, 32, 0, 13, 18, 20, 1, 5, 1, (0, 2)),
    (2, 64, 64, 32, 32, 64, 3, 3, 1, (1, 1)),    # cat[skip, up] 128 -> 64
    (1, 256, 0, 32, 32, 256, 3, 3, 1, (1, 1)),
    (3, 128, 0, 64, 64, 256, 3, 3, 2, (1, 1)),   # strided, tile rows not dividing
    (5, 480, 0, 8, 8, 480, 3, 3, 1, (1, 1)),     # 8x8 maps: 4 images per workgroup, ragged batch
    (19, 480, 0, 4, 4, 480, 3, 3, 1, (1, 1)),    # 4x4 maps: 8 images per workgroup, ragged batch
    (2, 30, 0, 16, 16, 30, 3, 3, 1, (1, 1)),     # 16x16, channels not multiples of 16
    (2, 256, 0, 1024, 1, 768, 1, 1, 1, (0, 0)),  # token projection [B,C,N,1]
    (1, 3072, 0, 1024, 1, 256, 1, 1, 1, (0, 0)), # FFN down projection
    (2, 32, 0, 50, 70, 4, 1, 1, 1, (0, 0)),      # seg head, ragged spatial
    (1, 81, 0, 64, 64, 64, 3, 3, 1, (1, 1)),     # cost-volume encoder, Cin = 81
    (2, 16, 8, 24, 40, 24, 3, 3, 1, (1, 1)),     # cat with ragged second input (C2 = 8 < chunk)
    (2, 32, 40, 16, 16, 48, 1, 1, 1, (0, 0)),    # 1x1 on cat, ragged second input
    (1, 6, 0, 33, 45, 70, 3, 3, 2, (1, 1)),      # stride 2, odd sizes, Cin < chunk, last sample's channel tail
]


@pytest.mark.parametrize("case", F16S_CASES)
def test_conv2d_f16s(dev, case):
    """f16-MFMA 3-term split kernel: fp32-class accuracy (same tolerance as the exact fp32 path)."""
    from cineflow import ops
    B, C1, C2, H, W, Cout, kh, kw, stride, pad = case
    x1 = randn(B, C1, H, W, seed=30)
    x2 = randn(B, C2, H, W, seed=31) if C2 else None
    w = randn(Cout, C1 + C2, kh, kw, seed=32) / math.sqrt((C1 + C2) * kh * kw)
    b = randn(Cout, seed=33)
    assert ops.f16s_supported(kh, kw, stride, pad) and ops.f16s_dynamic_ok(x1, x2, kh)
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    ref = F.conv2d(xin.double(), w.double(), b.double(), stride=stride, padding=pad)
    # a cat[x1, x2] whose split is not a chunk multiple is packed split-aware (x1's channels padded to whole chunks): same kernel
    wpk, ws = ops.pack_conv_weight_f16s(w.to(dev), c1=C1 if C2 else None)
    out = ops.conv2d_f16s(x1.to(dev), wpk, ws, b.to(dev), Cout, kh, kw, stride, pad, x2=None if x2 is None else x2.to(dev))
    check(out, ref, 1e-5, "conv_f16s")
    res = randn(*ref.shape, seed=34)
    big = torch.full((B, Cout + 3) + tuple(ref.shape[2:]), 7.0, device=dev)
    ops.conv2d_f16s(x1.to(dev), wpk, ws, b.to(dev), Cout, kh, kw, stride, pad, x2=None if x2 is None else x2.to(dev), act="gelu",
                    res=res.to(dev), out=big, out_coff=2)
    check(big[:, 2:2 + Cout], F.gelu(ref) + res, 1e-5, "conv_f16s epilogue")
    assert float(big[:, :2].min()) == 7.0 and float(big[:, 2 + Cout:].max()) == 7.0


def test_conv2d_f16s_batch_split(dev):
    """Inputs of 2 GiB and more (32-bit buffer offsets inside the kernel): the library cuts the batch into sub-batches itself --
    same kernel, same numbers as the per-sample calls, fused statistics and residual included."""
    from cineflow import ops
    B, C, H, W, Cout = 9, 64, 1024, 1024, 16         # 9 x 256 MiB = 2.25 GiB input
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, C, H, W, generator=g).to(dev)
    w = (torch.randn(Cout, C, 3, 3, generator=g) / math.sqrt(9 * C)).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    wpk, ws = ops.pack_conv_weight_f16s(w)
    assert x.numel() * 4 >= 2 ** 31
    out, st = ops.conv2d_f16s(x, wpk, ws, b, Cout, 3, 3, 1, (1, 1), act="relu", stats_groups=4)
    for i in (0, 4, 7, 8):                             # samples of both sub-batches (9 = 5 + 4: equal parts, see the next test)
        oi, si = ops.conv2d_f16s(x[i:i + 1], wpk, ws, b, Cout, 3, 3, 1, (1, 1), act="relu", stats_groups=4)
        assert torch.equal(out[i:i + 1], oi)
        assert torch.allclose(st.view(B, 4, 2)[i], si.view(4, 2), rtol=1e-7, atol=0)    # fp32 partial sums meet in another order
    ref = F.relu(F.conv2d(x[8:9].cpu().double(), w.cpu().double(), b.cpu().double(), padding=1))
    check(out[8:9], ref, 1e-5, "batch split")


def test_conv2d_f16s_prenorm_sub_batches_are_equal_parts(dev):
    """512 samples of 4 MiB: a descriptor holds 511, and a greedy cut would leave ONE sample, for which the dispatch picks the 8-wave
    shape of the small launches -- which has no deferred input normalisation although the capability probe (asked with 512) said yes.
    The batch is cut into equal parts (2 x 256) and the probe asks for the part sizes."""
    from cineflow import ops
    B, C, H = 512, 256, 64
    torch.manual_seed(5)
    raw = torch.randn(B, C, H, H, device=dev)
    assert ops.prenorm_ok(raw, C)
    w = (torch.randn(C, C, 3, 3, device=dev) / math.sqrt(9 * C))
    wpk, wsc = ops.pack_conv_weight_f16s(w)
    ws = torch.stack([raw.double().sum((2, 3)), (raw.double() ** 2).sum((2, 3))], dim=2).view(B, 8, C // 8, 2).sum(2).reshape(-1).contiguous()
    coef = ops.group_norm_coef(ws, None, None, 8, B, C, H * H)
    out = ops.conv2d_f16s_prenorm(raw, coef, -1.0, wpk, wsc, None, C)
    for lo, hi in ((0, 3), (253, 259), (509, 512)):      # across the cut at 256 and at both ends, against the two-pass route
        sl = raw[lo:hi].contiguous()
        assert not ops.prenorm_ok(sl, C)                  # a handful of samples: the small-launch shape, no deferred normalisation
        applied = ops.group_norm_apply(sl, None, None, 8, ws.view(B, -1)[lo:hi].reshape(-1).contiguous(), act="gelu")
        part = ops.conv2d_f16s(applied, wpk, wsc, None, C, 3, 3, 1, (1, 1))
        check(out[lo:hi], part.cpu(), 1e-5, "sub-batch %d:%d" % (lo, hi))
    ref = F.conv2d(F.gelu(F.group_norm(raw[511:512].cpu().double(), 8)), w.cpu().double(), padding=1)
    check(out[511:512], ref, 2e-5, "last sample vs fp64 reference")


def test_conv2d_f16s_nonfinite_inputs_propagate(dev):
    """NaN / Inf activations and values beyond fp16's range come out as NaN (loud), never as silently saturated numbers."""
    from cineflow import ops
    x = torch.zeros(1, 16, 8, 8)
    x[0, 3, 2, 2] = float("nan")
    x[0, 5, 6, 6] = 1e6
    w = torch.ones(16, 16, 3, 3) / 144
    wpk, ws = ops.pack_conv_weight_f16s(w.to(dev))
    out = ops.conv2d_f16s(x.to(dev), wpk, ws, None, 16, 3, 3, 1, (1, 1)).cpu()
    assert torch.isnan(out[0, :, 1:4, 1:4]).all() and torch.isnan(out[0, :, 5:8, 5:8]).all()
    assert torch.isfinite(out[0, :, 0, 4:]).all()


@pytest.mark.parametrize("C1,C2,H,W,k", [(6, 0, 16, 16, 3), (20, 12, 16, 16, 3), (20, 12, 13, 18, 3), (40, 24, 16, 16, 1), (30, 0, 64, 64, 3)])
def test_conv2d_f16s_channel_tail_never_reads_the_next_sample(dev, C1, C2, H, W, k):
    """the padded channel tail [C, chunk multiple) of sample b used to be fetched from the first channels of sample b + 1 (against
    zero weights): a NaN there became NaN * 0 in sample b.  The tail loads are parked out of range now: sample 0 of a batch whose sample 1 is
    all NaN / Inf equals sample 0 computed alone, bit for bit (vector, scalar and 1x1 staging; single input and cat[x1, x2])."""
    from cineflow import ops
    Cout = 24
    x1 = randn(2, C1, H, W, seed=60)
    x2 = randn(2, C2, H, W, seed=61) if C2 else None
    w = randn(Cout, C1 + C2, k, k, seed=62) / math.sqrt((C1 + C2) * k * k)
    wpk, ws = ops.pack_conv_weight_f16s(w.to(dev), c1=C1 if C2 else None)
    pad = (k // 2, k // 2)
    d = lambda t: None if t is None else t.to(dev)
    clean = ops.conv2d_f16s(d(x1[:1].contiguous()), wpk, ws, None, Cout, k, k, 1, pad, x2=d(None if x2 is None else x2[:1].contiguous()))
    x1[1] = float("nan")
    if x2 is not None:
        x2[1] = float("inf")
    both = ops.conv2d_f16s(d(x1), wpk, ws, None, Cout, k, k, 1, pad, x2=d(x2))
    assert torch.isfinite(both[0]).all(), "sample 0 poisoned by sample 1's values through the channel tail"
    assert torch.equal(both[0], clean[0])
    assert torch.isnan(both[1]).all()


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,stride,groups", [
    (2, 16, 32, 32, 64, 3, 1, 8),      # GroupNorm(8, 64): 8 channels per group
    (3, 32, 64, 64, 32, 3, 1, 32),     # InstanceNorm (one channel per group), narrow kernel variant
    (2, 64, 64, 64, 128, 3, 2, 8),     # stride 2, 16 channels per group
    (3, 256, 32, 32, 480, 3, 2, 480),  # stride 2 to 480 channels (128-channel stride-2 shape, padded last block), InstanceNorm
    (2, 48, 40, 56, 256, 1, 1, 8),     # 1x1, 32 channels per group, ragged spatial
    (5, 24, 8, 8, 96, 3, 1, 96),       # 8x8 maps: several samples per workgroup -> statistics pass fallback
    (2, 64, 33, 47, 512, 3, 1, 8),     # 64 channels per group: a group spans two m-tiles / workgroups
])
def test_conv_f16s_fused_group_norm_statistics(dev, B, Cin, H, W, Cout, k, stride, groups):
    from cineflow import ops
    x = randn(B, Cin, H, W, seed=45)
    w = randn(Cout, Cin, k, k, seed=46) / math.sqrt(Cin * k * k)
    b = randn(Cout, seed=47)
    g, be = 1 + 0.1 * randn(Cout, seed=48), 0.1 * randn(Cout, seed=49)
    y = F.conv2d(x, w, b, stride=stride, padding=k // 2)
    ref = F.gelu(F.group_norm(y, groups, g, be, 1e-5))
    wpk, ws_ = ops.pack_conv_weight_f16s(w.to(dev))
    out, stats = ops.conv2d_f16s(x.to(dev), wpk, ws_, b.to(dev), Cout, k, k, stride, (k // 2, k // 2), stats_groups=groups)
    check(out, y, 2e-5, "conv")
    cpg = Cout // groups
    yo = out.cpu().double()  # the statistics are those of the stored values
    want = torch.stack([yo.view(B, groups, -1).sum(-1), (yo ** 2).view(B, groups, -1).sum(-1)], -1)
    scale = yo.abs().view(B, groups, -1).sum(-1)[..., None] + 1.0
    rel = float(((stats.cpu().view(B, groups, 2) - want).abs() / scale).max())
    assert rel <= 2e-6, rel
    res = ops.group_norm_apply(out, g.to(dev), be.to(dev), groups, stats, act="gelu")
    check(res, ref, 3e-5, "gn apply on fused statistics")


@pytest.mark.parametrize("B,C1,C2,H,W,Cout,groups", [
    (12, 32, 0, 128, 128, 64, 8),      # 1536 tiles: more workgroups than fit on the chip at once
    (9, 16, 16, 128, 96, 128, 8),      # cat input, 128-channel (8-wave) workgroups
    (40, 16, 0, 64, 64, 32, 32),       # narrow variant, 1280 tiles, InstanceNorm statistics
    (11, 16, 0, 100, 132, 64, 8),      # ragged tiles (100 rows, 132 columns)
    (70, 48, 48, 32, 32, 480, 480),    # cat input -> 480 channels (padded to 4 x 128-channel blocks), InstanceNorm statistics, 1120 workgroups
    (6, 64, 0, 64, 64, 480, 8),        # 60 channels per group: groups straddle the 128-channel blocks, the last block ends at channel 480
])
def test_conv_f16s_many_tiles(dev, B, C1, C2, H, W, Cout, groups):
    """Launches with more tiles than resident workgroups (several dispatch rounds, XCD-banded tile order, vector staging at the
    image borders): values and fused GroupNorm statistics against torch."""
    from cineflow import ops
    x1 = randn(B, C1, H, W, seed=70)
    x2 = randn(B, C2, H, W, seed=71) if C2 else None
    w = randn(Cout, C1 + C2, 3, 3, seed=72) / math.sqrt((C1 + C2) * 9)
    b = randn(Cout, seed=73)
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    y = F.conv2d(xin, w, b, padding=1)
    wpk, ws_ = ops.pack_conv_weight_f16s(w.to(dev))
    out, stats = ops.conv2d_f16s(x1.to(dev), wpk, ws_, b.to(dev), Cout, 3, 3, 1, (1, 1), x2=None if x2 is None else x2.to(dev),
                                 stats_groups=groups)
    check(out, y, 3e-5, "conv")
    yo = out.cpu().double()
    want = torch.stack([yo.view(B, groups, -1).sum(-1), (yo ** 2).view(B, groups, -1).sum(-1)], -1)
    scale = yo.abs().view(B, groups, -1).sum(-1)[..., None] + 1.0
    rel = float(((stats.cpu().view(B, groups, 2) - want).abs() / scale).max())
    assert rel <= 2e-6, rel
    # without statistics and without bias, into a channel slice of a wider tensor
    big = torch.zeros(B, Cout + 8, H, W, device=dev)
    ops.conv2d_f16s(x1.to(dev), wpk, ws_, None, Cout, 3, 3, 1, (1, 1), x2=None if x2 is None else x2.to(dev), out=big, out_coff=8)
    check(big[:, 8:], F.conv2d(xin, w, None, padding=1), 3e-5, "conv, channel slice")
    assert float(big[:, :8].abs().max()) == 0.0


@pytest.mark.parametrize("B,C,H,W,Cout", [
    (3, 32, 64, 64, 32),       # narrow shape (Generic_UNet level 0)
    (2, 64, 48, 64, 64),       # 64-channel shape, ragged rows
    (32, 128, 64, 64, 128),    # 128-channel four-wave shape (>= 1024 workgroups)
    (2, 480, 16, 16, 480),     # channel tail (480 = 30 chunks), 16-wide rows
    (140, 480, 16, 16, 480),   # 480 = 3 x 128 + 96 output channels on the four-wave 128-channel shape (>= 1024 workgroups), last block 3/4 full
    (3, 224, 32, 32, 224),     # 224 = 128 + 96 on the eight-wave 128-channel shape
])
def test_conv_f16s_prenorm(dev, B, C, H, W, Cout):
    """convolution that applies its input's deferred InstanceNorm + LeakyReLU while staging, against torch on the materialised
    activation; fused output statistics; the capability query"""
    from cineflow import ops
    x = randn(B, C, H, W, seed=100) * 1.7 + 0.4
    g, bt = randn(C, seed=101), randn(C, seed=102)
    w = randn(Cout, C, 3, 3, seed=103) / math.sqrt(C * 9)
    b = randn(Cout, seed=104)
    act = F.leaky_relu(F.instance_norm(x, weight=g, bias=bt, eps=1e-5), 0.01)
    want = F.conv2d(act, w, b, padding=1)
    xd = x.to(dev)
    assert ops.prenorm_ok(xd, Cout)
    xs = x.double().view(B, C, -1)
    ws = torch.stack([xs.sum(-1), (xs ** 2).sum(-1)], -1).reshape(-1).to(dev)
    coef = ops.group_norm_coef(ws, g.to(dev), bt.to(dev), C, B, C, H * W)
    check(coef[:, 0], x.view(B, C, -1).mean(-1), 1e-6, "coef mean")
    wpk, wsc = ops.pack_conv_weight_f16s(w.to(dev))
    out, st = ops.conv2d_f16s_prenorm(xd, coef, 0.01, wpk, wsc, b.to(dev), Cout, stats_groups=Cout)
    check(out, want, 3e-5, "prenorm conv")
    yo = out.cpu().double().view(B, Cout, -1)
    wst = torch.stack([yo.sum(-1), (yo ** 2).sum(-1)], -1)
    assert float(((st.cpu().view(B, Cout, 2) - wst).abs() / (yo.abs().sum(-1)[..., None] + 1.0)).max()) <= 2e-6
    # identical to the two-kernel path on the same operands
    two = ops.conv2d_f16s(ops.group_norm_apply(xd, g.to(dev), bt.to(dev), C, ws, act="lrelu", out=torch.empty_like(xd)), wpk, wsc, b.to(dev), Cout, 3, 3, 1,
                          (1, 1))
    check(out, two.cpu(), 2e-5, "prenorm vs apply + conv")
    # the DoubleConv form: GroupNorm(8) + GELU (in_slope < 0) deferred to the consumer
    xg = x.double().view(B, 8, -1)
    wsg = torch.stack([xg.sum(-1), (xg ** 2).sum(-1)], -1).reshape(-1).to(dev)
    coefg = ops.group_norm_coef(wsg, g.to(dev), bt.to(dev), 8, B, C, H * W)
    wantg = F.conv2d(F.gelu(F.group_norm(x, 8, g, bt, eps=1e-5)), w, b, padding=1)
    check(ops.conv2d_f16s_prenorm(xd, coefg, -1.0, wpk, wsc, b.to(dev), Cout), wantg, 3e-5, "prenorm conv, GroupNorm + GELU")
    assert not ops.prenorm_ok(torch.empty(2, 32, 30, 30, device=dev), 32)          # W % 4 != 0: scalar staging
    assert not ops.prenorm_ok(torch.empty(4, 480, 8, 8, device=dev), 480)          # several samples per workgroup
    assert not ops.prenorm_ok(torch.empty(2, 128, 32, 32, device=dev), 128)        # < 1024 workgroups: the 8-wave shape keeps the apply pass
    with pytest.raises(RuntimeError):
        ops.conv2d_f16s_prenorm(torch.zeros(2, 32, 30, 30, device=dev), coef[:2, :, :32].contiguous(), 0.01, wpk, wsc, None, Cout)


@pytest.mark.parametrize("B,C,H,W", [(2, 64, 64, 64), (3, 32, 16, 16), (2, 16, 9, 7)])
@pytest.mark.parametrize("mode", ["after_act", "before_act"])
def test_group_norm_apply_res_norm(dev, B, C, H, W, mode):
    """apply pass whose residual is normalised on the fly (DoubleConv's 1x1 conv + GroupNorm branch) against torch and against
    the two-pass form; large and small plane kernels, vector and scalar paths"""
    from cineflow import ops
    G = 8
    x, r = randn(B, C, H, W, seed=90) * 2 + 0.3, randn(B, C, H, W, seed=91) * 3 - 1.0
    g1, b1, g2, b2 = randn(C, seed=92), randn(C, seed=93), randn(C, seed=94), randn(C, seed=95)
    rn = F.group_norm(r, G, g2, b2, eps=1e-5)
    y = F.group_norm(x, G, g1, b1, eps=1e-5)
    want = F.gelu(y + rn) if mode == "before_act" else F.gelu(y) + rn

    def stats(t):
        td = t.double().view(B, G, -1)
        return torch.stack([td.sum(-1), (td ** 2).sum(-1)], -1).reshape(-1).to(dev)
    xd, rd = x.to(dev), r.to(dev)
    got = ops.group_norm_apply(xd, g1.to(dev), b1.to(dev), G, stats(x), act="gelu", res=rd, res_mode=mode,
                               res_norm=(stats(r), g2.to(dev), b2.to(dev)))
    check(got, want, 1e-5, "fused residual norm")
    two = ops.group_norm_apply(xd, g1.to(dev), b1.to(dev), G, stats(x), act="gelu", res_mode=mode,
                               res=ops.group_norm_apply(rd, g2.to(dev), b2.to(dev), G, stats(r)))
    check(got, two.cpu(), 2e-6, "fused vs two passes")


@pytest.mark.parametrize("B,Cin,H,W,Cout,K,groups", [
    (3, 1, 64, 64, 32, 3, 32),        # Generic_UNet stem, InstanceNorm statistics (one group per channel)
    (2, 1, 256, 256, 64, 3, 8),       # flow encoder stem, GroupNorm(8)
    (2, 6, 70, 100, 64, 3, 8),        # ragged: 70 rows (not a multiple of the 8-row blocks), 100 columns (not a multiple of 64)
    (2, 6, 64, 64, 64, 1, 8),         # 1x1 downsample branch of the stem
    (1, 2, 33, 17, 16, 3, None),      # two channels, no statistics, image smaller than a block
    (2, 1, 40, 72, 24, 1, 4),
])
def test_conv_small_cin(dev, B, Cin, H, W, Cout, K, groups):
    """cf_conv2d_small_cin (direct fp32 stem convolution) and its fused statistics against torch; exact fp32 FMA chain, so the
    tolerance is fp32 summation order on <= 54 terms of O(1) products (5e-6 on outputs up to ~5)."""
    from cineflow import ops
    x = randn(B, Cin, H, W, seed=80)
    w = randn(Cout, Cin, K, K, seed=81) / math.sqrt(Cin * K * K)
    b = randn(Cout, seed=82)
    y = F.conv2d(x, w, b, padding=K // 2)
    assert ops.small_cin_supported(Cin, K, K, 1, (K // 2, K // 2), groups) == (not (Cin == 6 and K == 3))
    r = ops.conv2d_small_cin(x.to(dev), w.to(dev), b.to(dev), groups)
    out, stats = r if groups else (r, None)
    check(out, y, 5e-6, "direct conv")
    if groups:
        yo = out.cpu().double()
        want = torch.stack([yo.view(B, groups, -1).sum(-1), (yo ** 2).view(B, groups, -1).sum(-1)], -1)
        scale = yo.abs().view(B, groups, -1).sum(-1)[..., None] + 1.0
        assert float(((stats.cpu().view(B, groups, 2) - want).abs() / scale).max()) <= 2e-6
    check(ops.conv2d_small_cin(x.to(dev), w.to(dev), None, None), F.conv2d(x, w, None, padding=K // 2), 5e-6, "direct conv, no bias")
    assert not ops.small_cin_supported(3, 3, 3, 1, (1, 1)) and not ops.small_cin_supported(1, 3, 3, 2, (1, 1))
    with pytest.raises(RuntimeError):
        ops.conv2d_small_cin(randn(1, 3, 8, 8, seed=1).to(dev), randn(4, 3, 3, 3, seed=2).to(dev), None, None)


def test_conv_f16s_dynamic_range(dev):
    """tiny and large operands: the weight pre-scaling keeps the lo halves normal; activations lose <= 2^-25 absolute."""
    from cineflow import ops
    for ax, aw in ((1.0, 1e-4), (20.0, 0.01), (0.05, 0.06), (300.0, 1e-3)):
        x, w = ax * randn(1, 64, 32, 32, seed=43), aw * randn(64, 64, 3, 3, seed=44)
        ref = F.conv2d(x.double(), w.double(), padding=1)
        wpk, ws = ops.pack_conv_weight_f16s(w.to(dev))
        out = ops.conv2d_f16s(x.to(dev), wpk, ws, None, 64, 3, 3, 1, (1, 1))
        rel = maxdiff(out, ref) / float(ref.abs().max())
        assert rel <= 2e-6, (ax, aw, rel)


def test_conv_transpose_f16s(dev):
    from cineflow import ops
    for (B, Cin, H, W, Cout, bias) in [(2, 32, 8, 8, 16, True), (1, 20, 5, 7, 9, False), (1, 256, 32, 32, 64, True), (3, 480, 4, 4, 480, False)]:
        x, w = randn(B, Cin, H, W, seed=37), randn(Cin, Cout, 2, 2, seed=38) / math.sqrt(Cin)
        b = randn(Cout, seed=39) if bias else None
        ref = F.conv_transpose2d(x.double(), w.double(), None if b is None else b.double(), stride=2)
        wpk, ws = ops.pack_conv_weight_f16s(w.to(dev).permute(1, 2, 3, 0).reshape(Cout * 4, Cin, 1, 1))
        out = ops.conv_transpose2d_k2s2_f16s(x.to(dev), wpk, ws, None if b is None else b.to(dev), Cout)
        check(out, ref, 1e-5, "convT f16s")
        # GroupNorm statistics of the scattered output from the epilogue (one sample per workgroup) or the statistics pass (small maps)
        for groups in ((8, Cout) if Cout % 8 == 0 else (Cout,)):
            out2, st = ops.conv_transpose2d_k2s2_f16s(x.to(dev), wpk, ws, None if b is None else b.to(dev), Cout, stats_groups=groups)
            assert torch.equal(out2, out)
            yo = out.cpu().double()
            want = torch.stack([yo.view(B, groups, -1).sum(-1), (yo ** 2).view(B, groups, -1).sum(-1)], -1)
            scale = yo.abs().view(B, groups, -1).sum(-1)[..., None] + 1.0
            assert float(((st.cpu().view(B, groups, 2) - want).abs() / scale).max()) <= 2e-6, (B, Cin, H, W, Cout, groups)


@pytest.mark.parametrize("act", ["relu", "lrelu", "tanh", "sigmoid"])
def test_conv_activations(dev, act):
    from cineflow import ops
    x, w = randn(1, 8, 9, 9, seed=35), randn(12, 8, 3, 3, seed=36) / 8
    ref = F.conv2d(x, w, None, padding=1)
    fn = {"relu": F.relu, "lrelu": lambda t: F.leaky_relu(t, 0.01), "tanh": torch.tanh, "sigmoid": torch.sigmoid}[act]
    out = ops.conv2d(x.to(dev), ops.prep_conv_weight(w).to(dev), None, 12, 3, 3, 1, (1, 1), act=act)
    check(out, fn(ref), 1e-5)


def test_conv_transpose(dev):
    from cineflow import ops
    for (B, Cin, H, W, Cout, bias) in [(2, 32, 8, 8, 16, True), (1, 20, 5, 7, 9, False), (1, 256, 32, 32, 64, True)]:
        x, w = randn(B, Cin, H, W, seed=37), randn(Cin, Cout, 2, 2, seed=38) / math.sqrt(Cin)
        b = randn(Cout, seed=39) if bias else None
        ref = F.conv_transpose2d(x, w, b, stride=2)
        out = ops.conv_transpose2d_k2s2(x.to(dev), w.to(dev), None if b is None else b.to(dev))
        check(out, ref, 2e-5, "convT")


def test_conv_full_size_linearity(dev):
    """BASELINE-size layer (64->64 @ 256x256): conv(a+b) == conv(a)+conv(b), and a centre crop matches the CPU."""
    from cineflow import ops
    w = randn(64, 64, 3, 3, seed=40) / 24
    wt = ops.prep_conv_weight(w).to(dev)
    a, b = randn(2, 64, 256, 256, seed=41).to(dev), randn(2, 64, 256, 256, seed=42).to(dev)
    ya, yb = ops.conv2d(a, wt, None, 64, 3, 3, 1, (1, 1)), ops.conv2d(b, wt, None, 64, 3, 3, 1, (1, 1))
    yab = ops.conv2d(ops.add(a, b), wt, None, 64, 3, 3, 1, (1, 1))
    check(yab, ops.add(ya, yb).cpu(), 2e-4)
    ref = F.conv2d(a[:1, :, 96:160, 96:160].cpu(), w, None, padding=1)
    check(ya[:1, :, 97:159, 97:159], ref[:, :, 1:-1, 1:-1], 5e-5)


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("B,C,H,W,groups", [(2, 16, 32, 32, 8), (1, 64, 64, 64, 8), (3, 8, 5, 7, 8), (2, 24, 16, 16, 24), (2, 480, 4, 4, 480),
                                            (1, 64, 256, 256, 8)])
def test_group_norm(dev, B, C, H, W, groups):
    from cineflow import ops
    x = 3 * randn(B, C, H, W, seed=50) + 1.5
    g, b = 1 + 0.1 * randn(C, seed=51), 0.1 * randn(C, seed=52)
    ref = F.group_norm(x, groups, g, b, 1e-5)
    check(ops.group_norm(x.to(dev), g.to(dev), b.to(dev), groups), ref, 2e-5, "gn")
    res = randn(B, C, H, W, seed=53)
    check(ops.group_norm(x.to(dev), g.to(dev), b.to(dev), groups, act="gelu", res=res.to(dev), res_mode="after_act"), F.gelu(ref) + res, 2e-5)
    check(ops.group_norm(x.to(dev), g.to(dev), b.to(dev), groups, act="gelu", res=res.to(dev), res_mode="before_act"), F.gelu(ref + res), 2e-5)
    check(ops.group_norm(x.to(dev), g.to(dev), b.to(dev), groups, act="lrelu"), F.leaky_relu(ref, 0.01), 2e-5)


def test_instance_norm_matches_torch(dev):
    from cineflow import ops
    x = randn(2, 12, 9, 11, seed=54)
    g, b = 1 + 0.1 * randn(12, seed=55), 0.1 * randn(12, seed=56)
    check(ops.group_norm(x.to(dev), g.to(dev), b.to(dev), 12), F.instance_norm(x, weight=g, bias=b, eps=1e-5), 2e-5)


def test_zscore_whole_block(dev):
    from cineflow.inference import normalize_intensity_
    from oracle import ops as OO
    x = 50 * randn(5, 1, 32, 32, seed=57) + 300
    check(normalize_intensity_(x.clone().to(dev)), OO.normalize_intensity(x), 2e-5)


@pytest.mark.parametrize("B,C,N", [(3, 32, 64), (2, 256, 1024), (1, 512, 100), (2, 7, 33), (1, 300, 70)])
def test_layer_norm_cf(dev, B, C, N):
    """register-resident path (C <= 256), re-read path (C > 256), ragged token counts and channel counts"""
    from cineflow import ops
    x = randn(B, C, N, seed=58) * 2 + 0.5
    g, b = 1 + 0.1 * randn(C, seed=59), 0.1 * randn(C, seed=60)
    ref = F.layer_norm(x.permute(0, 2, 1), (C,), g, b, 1e-5).permute(0, 2, 1)
    check(ops.layer_norm_cf(x.to(dev), g.to(dev), b.to(dev)), ref, 2e-5)


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,heads,d,Nq,Nk", [(2, 4, 8, 64, 64), (1, 8, 8, 64, 64), (2, 4, 64, 256, 256), (1, 4, 64, 1024, 1024), (1, 2, 16, 32, 96),
                                             (1, 2, 32, 160, 64)])
def test_attention(dev, B, heads, d, Nq, Nk):
    from cineflow import ops
    C = heads * d
    q, k, v = randn(B, C, Nq, seed=61), randn(B, C, Nk, seed=62), randn(B, C, Nk, seed=63)
    qh = q.view(B, heads, d, Nq).permute(0, 1, 3, 2)
    kh = k.view(B, heads, d, Nk).permute(0, 1, 3, 2)
    vh = v.view(B, heads, d, Nk).permute(0, 1, 3, 2)
    att = torch.softmax((qh / math.sqrt(d)) @ kh.transpose(-1, -2), dim=-1) @ vh
    ref = att.permute(0, 1, 3, 2).reshape(B, C, Nq)
    check(ops.attention_cf(q.to(dev), k.to(dev), v.to(dev), heads), ref, 2e-5, "attention")


def test_attention_slices_and_extremes(dev):
    from cineflow import ops
    B, heads, d, N = 2, 4, 8, 64
    C = heads * d
    qkv = randn(B, 3 * C, N, seed=64).to(dev)
    q, k, v = qkv.narrow(1, 0, C), qkv.narrow(1, C, C), qkv.narrow(1, 2 * C, C)
    a = ops.attention_cf(q, k, v, heads)
    b = ops.attention_cf(q.contiguous(), k.contiguous(), v.contiguous(), heads)
    check(a, b.cpu(), 0)
    # large logits (online-softmax rescale path): one key dominates each query
    big = qkv.clone()
    big[:, :2 * C] *= 30.0
    out = ops.attention_cf(big.narrow(1, 0, C), big.narrow(1, C, C), big.narrow(1, 2 * C, C), heads)
    qh = big[:, :C].cpu().view(B, heads, d, N).permute(0, 1, 3, 2)
    kh = big[:, C:2 * C].cpu().view(B, heads, d, N).permute(0, 1, 3, 2)
    vh = big[:, 2 * C:].cpu().view(B, heads, d, N).permute(0, 1, 3, 2)
    ref = (torch.softmax((qh.double() / math.sqrt(d)) @ kh.double().transpose(-1, -2), -1) @ vh.double()).permute(0, 1, 3, 2).reshape(B, C, N)
    check(out, ref, 5e-4)
    assert torch.isfinite(out).all()
    # constant V -> output equals V whatever the scores
    vc = torch.ones(B, C, N, device=dev) * 2.5
    check(ops.attention_cf(q, k, vc, heads), vc.cpu(), 1e-5)


# ------------------------------------------------------------------------------------------------ plumbing kernels
def test_gru_and_elementwise(dev):
    from cineflow import ops
    B, C, HW = 2, 8, 20
    gates, h, cand = torch.sigmoid(randn(B, 2 * C, HW, seed=70)), randn(B, C, HW, seed=71), torch.tanh(randn(B, C, HW, seed=72))
    check(ops.gru_reset_mul(gates.to(dev), h.to(dev)), gates[:, :C] * h, 1e-7)
    u = gates[:, C:]
    check(ops.gru_blend(gates.to(dev), h.to(dev), cand.to(dev)), (1 - u) * h + u * cand, 1e-6)
    a, b = randn(3, 5, 7, seed=73), randn(5, 7, seed=74)
    check(ops.add(a.to(dev), b.to(dev)), a + b, 0)
    check(ops.sub(a.to(dev), a.to(dev)), torch.zeros_like(a), 0)
    check(ops.mul(a.to(dev), b.to(dev)), a * b, 0)
    src = randn(2, 6, 4, 5, seed=75)
    dst = torch.zeros(2, 9, 4, 5, device=dev)
    ops.copy_channels(src.to(dev), 1, 3, dst=dst, dst_coff=4, act="relu")
    check(dst[:, 4:7], F.relu(src[:, 1:4]), 0)
    assert float(dst[:, :4].abs().sum()) == 0 and float(dst[:, 7:].abs().sum()) == 0


def test_crop_pad_flip(dev):
    from cineflow import ops
    x = randn(3, 2, 20, 30, seed=76)
    c = ops.crop2d(x.to(dev), 3, 5, 8, 16)
    check(c, x[..., 3:11, 5:21], 0)
    p = ops.pad2d(c, 3, 5, 20, 30)
    ref = torch.zeros_like(x)
    ref[..., 3:11, 5:21] = x[..., 3:11, 5:21]
    check(p, ref, 0)
    check(ops.flip2d(x.to(dev), 1, 0), torch.flip(x, (2,)), 0)
    check(ops.flip2d(x.to(dev), 0, 1), torch.flip(x, (3,)), 0)
    check(ops.flip2d(x.to(dev), 1, 1), torch.flip(x, (2, 3)), 0)


def test_processor_device_roundtrip(dev):
    from cineflow.inference import Processor
    from oracle.models import Processor as OP
    data = randn(4, 1, 40, 40, seed=77)
    for c in [(20, 20), (2, 3), (39, 39)]:
        p, o = Processor(16, 40), OP(16, 40)
        crop, pad = p.crop_and_pad(data.to(dev), c)
        ocrop, opad = o.crop_and_pad(data, c)
        check(crop, ocrop, 0)
        back = p.uncrop_no_registration(crop, pad)
        check(back, o.uncrop_no_registration(ocrop[None], opad[None])[0], 0)


def test_tta_and_tiles(dev):
    from cineflow import ops
    B, K, H, W = 2, 4, 12, 10
    logits = randn(B, K, H, W, seed=78)
    acc = torch.zeros(B, K, H, W, device=dev)
    ops.tta_accumulate(logits.to(dev), acc, 0, 0, 0.25)
    ops.tta_accumulate(torch.flip(logits, (3,)).contiguous().to(dev), acc, 0, 1, 0.25)
    ops.tta_accumulate(torch.flip(logits, (2,)).contiguous().to(dev), acc, 1, 0, 0.25)
    ops.tta_accumulate(torch.flip(logits, (2, 3)).contiguous().to(dev), acc, 1, 1, 0.25)
    check(acc, torch.softmax(logits, 1), 1e-6)
    agg, cnt = torch.zeros(K, 20, 18, device=dev), torch.zeros(K, 20, 18, device=dev)
    g = torch.rand(12, 10, generator=torch.Generator().manual_seed(79)) + 0.1
    pred = torch.softmax(logits[0], 0) * g
    ops.tile_accumulate(pred.to(dev), g.to(dev), agg, cnt, 3, 4)
    ops.tile_accumulate(pred.to(dev), g.to(dev), agg, cnt, 8, 8)
    ra, rc = torch.zeros(K, 20, 18), torch.zeros(K, 20, 18)
    for (lx, ly) in ((3, 4), (8, 8)):
        ra[:, lx:lx + 12, ly:ly + 10] += pred
        rc[:, lx:lx + 12, ly:ly + 10] += g
    check(agg, ra, 1e-6)
    check(cnt, rc, 1e-6)
    cnt2 = cnt.clamp(min=1e-3)
    seg, probs = ops.tile_finalize(agg, cnt2)
    check(probs, ra / rc.clamp(min=1e-3), 1e-5)
    assert torch.equal(seg.cpu().long(), (ra / rc.clamp(min=1e-3)).argmax(0))
    am = ops.argmax_channels(logits.to(dev))
    assert torch.equal(am.cpu().long(), logits.argmax(1))


# ------------------------------------------------------------------------------------------------ export post-processing
def test_connected_component_filter(dev, golden):
    """Largest-connected-component filter (connected_components.py:51-107) on the device vs the reference's outputs."""
    from cineflow import ops
    from oracle import ops as OO
    g = golden("connected_components")
    cases = [([1, 2, 3], None), ([(1, 2), 3], None), ([1, 2], {1: 40.0, 2: 1e9}), (None, None)]
    for i, (fw, mv) in enumerate(cases):
        img = torch.from_numpy(g["labels"].copy()).to(dev)
        out, largest_removed, kept = ops.remove_all_but_the_largest_connected_component(img, fw, 1.5, mv)
        assert (out.cpu().numpy() == g["out%d" % i]).all(), "case %d" % i
        _, lr_o, ks_o = OO.remove_all_but_the_largest_connected_component(g["labels"].copy(), fw, 1.5, mv)
        assert kept == ks_o and largest_removed == lr_o, (kept, ks_o, largest_removed, lr_o)
    # 2-D, a long snake (many sweeps) next to a small island, and an empty class
    img = np.zeros((64, 96), np.uint8)
    for r in range(0, 64, 4):
        img[r, 2:94] = 1
        img[r:r + 4, 93 if (r // 4) % 2 == 0 else 2] = 1
    img[60:63, 40:44] = 0
    img[62, 50:53] = 1
    ref = OO.remove_all_but_the_largest_connected_component(img.copy(), [1, 2], 1.0, None)[0]
    out = ops.remove_all_but_the_largest_connected_component(torch.from_numpy(img.copy()).to(dev), [1, 2], 1.0, None)[0]
    assert (out.cpu().numpy() == ref).all()


@pytest.mark.parametrize("shape,new,order,sep,axis,oz", [((5, 20, 24), (5, 31, 40), 1, True, [0], 0), ((5, 20, 24), (8, 31, 17), 1, True, [0], 0),
                                                          ((6, 20, 24), (9, 15, 40), 1, False, None, 0), ((4, 33, 21), (4, 50, 50), 0, False, None, 0),
                                                          ((5, 20, 24), (7, 30, 30), 1, True, [0], 1)])
def test_export_resampling(dev, shape, new, order, sep, axis, oz):
    """resample_data_or_seg (preprocessing.py:111-200) for the orders the export uses, against the numpy/scipy restatement
    (skimage and batchgenerators are absent: parity unpinned)."""
    from cineflow import ops
    from oracle import ops as OO
    rng = np.random.RandomState(3)
    data = rng.rand(3, *shape).astype(np.float32)
    ref = OO.resample_data_or_seg(data, new, False, axis, order, sep, oz)
    out = ops.resample_data_or_seg(data, new, False, axis, order, sep, oz)
    assert out.shape == ref.shape == (3,) + tuple(new) and out.dtype == np.float32
    check(torch.from_numpy(out), ref, 2e-6, "data order %d" % order)
    seg = rng.randint(0, 4, (1,) + tuple(shape)).astype(np.uint8)
    ref = OO.resample_data_or_seg(seg, new, True, axis, 0, sep, 0)
    out = ops.resample_data_or_seg(seg, new, True, axis, 0, sep, 0)
    assert out.dtype == np.uint8 and np.array_equal(out, ref)
    # identity and constant fields
    assert ops.resample_data_or_seg(data, shape, False, axis, order, sep, oz) is data
    const = np.full((1,) + tuple(shape), 2.5, np.float32)
    check(torch.from_numpy(ops.resample_data_or_seg(const, new, False, axis, order, sep, oz)), np.full((1,) + tuple(new), 2.5), 1e-6)

def test_norm_head_1x1_vs_separate_passes(dev):
    """cf_norm_head_1x1 (deferred InstanceNorm + LeakyReLU folded into Generic_UNet's 1x1 head, generic_UNet.py:405-408) against the
    oracle's two steps -- F.instance_norm(affine) + leaky_relu, then a bias-free 1x1 convolution -- and against the device's own
    apply-pass + 1x1 route; 32 channels x 4 classes at 256x256 (the bench U-Net's head) and an odd shape."""
    from cineflow import ops
    for (B, C, H, W, K, with_bias) in [(3, 32, 256, 256, 4, False), (2, 20, 12, 36, 2, True), (1, 64, 64, 64, 8, False)]:
        raw = (2.0 * randn(B, C, H, W, seed=C) + randn(1, C, 1, 1, seed=K)).to(dev)
        gamma, beta = (1.0 + 0.2 * randn(C, seed=1)).to(dev), (0.1 * randn(C, seed=2)).to(dev)
        w = (randn(K, C, 1, 1, seed=3) / math.sqrt(C)).to(dev)
        bias = (0.3 * randn(K, seed=4)).to(dev) if with_bias else None
        ref = F.conv2d(F.leaky_relu(F.instance_norm(raw.cpu().double(), weight=gamma.cpu().double(), bias=beta.cpu().double(), eps=1e-5), 0.01),
                       w.cpu().double(), None if bias is None else bias.cpu().double())
        # statistics exactly as the producing convolution would leave them: fp64 {sum, sum of squares} per (sample, channel)
        ws = torch.stack([raw.double().sum((2, 3)), (raw.double() ** 2).sum((2, 3))], dim=2).reshape(-1).contiguous()
        coef = ops.group_norm_coef(ws, gamma, beta, C, B, C, H * W, 1e-5)
        assert ops.norm_head_ok(raw, K)
        out = ops.norm_head_1x1(raw, coef, 0.01, w, bias)
        assert out.shape == (B, K, H, W)
        check(out, ref, 2e-5, "norm_head %dx%dx%d" % (C, H, W))
        applied = ops.group_norm_apply(raw, gamma, beta, C, ws, eps=1e-5, act="lrelu")
        two_pass = F.conv2d(applied.cpu().double(), w.cpu().double(), None if bias is None else bias.cpu().double())
        check(out, two_pass, 2e-5, "norm_head vs apply + 1x1")
    assert not ops.norm_head_ok(torch.zeros(1, 4, 3, 3, device=dev), 4) and not ops.norm_head_ok(torch.zeros(1, 4, 4, 4, device=dev), 3)


def test_f16s_range_check_counts_unrepresentable_inputs(dev):
    """the debug-mode counter for f16-split convolution inputs outside the supported range (|x| >= 65504, NaN, Inf)"""
    from cineflow import ops
    x = torch.zeros(2, 16, 8, 8)
    x[0, 1, 2, 3], x[1, 5, 0, 0], x[1, 6, 7, 7], x[0, 0, 0, 0] = 7e4, float("nan"), float("-inf"), 65000.0     # three bad, one large but fine
    w = torch.ones(16, 16, 3, 3) / 144
    wpk, ws = ops.pack_conv_weight_f16s(w.to(dev))
    old = ops.F16S_RANGE_CHECK
    try:
        ops.F16S_RANGE_CHECK = True
        ops.f16s_range_violations(reset=True)
        ops.conv2d_f16s(x.to(dev), wpk, ws, None, 16, 3, 3, 1, (1, 1))
        assert ops.f16s_range_violations() == 3
        ops.conv2d_f16s(torch.randn(2, 16, 8, 8).to(dev), wpk, ws, None, 16, 3, 3, 1, (1, 1))
        assert ops.f16s_range_violations(reset=True) == 3 and ops.f16s_range_violations() == 0
    finally:
        ops.F16S_RANGE_CHECK = old


@pytest.mark.parametrize("B,Cin,H,W,Cout,bias,res", [(2, 64, 256, 256, 2, True, False),    # Decoder2D.final_conv at full size
                                                      (3, 256, 32, 32, 2, True, True),      # RAFT FlowHead.conv2 with the residual input
                                                      (1, 5, 13, 70, 4, False, False),      # ragged: partial column block, partial row block
                                                      (2, 16, 9, 64, 1, True, True), (1, 8, 40, 130, 3, False, True)])
def test_conv_small_cout_direct(dev, B, Cin, H, W, Cout, bias, res):
    """cf_conv2d_small_cout: the flow heads as a direct exact-fp32 kernel (buffer-resource taps: zero padding by range check)"""
    from cineflow import ops
    from cineflow.nn import Conv2d
    x = randn(B, Cin, H, W, seed=70)
    w = randn(Cout, Cin, 3, 3, seed=71) / math.sqrt(9 * Cin)
    b = randn(Cout, seed=72) if bias else None
    r = randn(B, Cout, H, W, seed=73) if res else None
    ref = F.conv2d(x.double(), w.double(), None if b is None else b.double(), padding=1) + (0 if r is None else r.double())
    out = ops.conv2d_small_cout(x.to(dev), w.to(dev), None if b is None else b.to(dev), None if r is None else r.to(dev))
    check(out, ref, 2e-5, "small_cout")          # fp32 accumulation over 9 * Cin terms
    assert ops.small_cout_supported(Cout, 3, 3, 1, (1, 1)) and not ops.small_cout_supported(Cout, 3, 3, 2, (1, 1)) and not ops.small_cout_supported(8, 3, 3, 1, (1, 1))
    m = Conv2d(Cin, Cout, 3, padding=1, bias=bias)                              # the module routes there by itself
    m.load_state_dict({"weight": w, **({"bias": b} if bias else {})}, dev)
    y = m(x.to(dev), res=None if r is None else r.to(dev))
    assert torch.equal(y, out)


# ------------------------------------------------------------------------------------------------ conv_stream.hip (persistent pipelined kernel)
STREAM_CASES = [   # B, C1, C2, H, W, Cout, groups
    (20, 64, 0, 128, 128, 64, 8),      # 1280 tiles = 5 per workgroup (odd), 4 chunks
    (32, 32, 0, 128, 128, 32, 32),     # 1024 tiles (16 rows each) = 4 per workgroup, narrow (one m-tile), 2 chunks, InstanceNorm statistics
    (34, 32, 32, 128, 128, 32, 32),    # cat input, narrow
    (33, 32, 0, 120, 132, 32, 32),     # narrow, ragged rows (120 = 7.5 tiles) and columns
    (12, 64, 64, 128, 160, 64, 8),     # cat input, 5 tile columns
    (12, 81, 0, 128, 128, 64, 8),      # 81 channels: 6 chunks, zero-weight channel tail
    (10, 20, 28, 100, 132, 64, 8),     # split-aware packing (C1 = 20 padded to 2 chunks), ragged rows (100 = 12.5 tiles) and columns (132)
    (3, 64, 0, 256, 256, 64, 8),       # 768 tiles: below the threshold -> stays on conv_f16s (the knob must change nothing)
    # the bench's map size with the kernel engaged (256 x 256: tiles_x = 8; >= 1024 tiles): the shapes that carry ~16 % of the bench step
    (8, 32, 0, 256, 256, 32, 32),      # Generic_UNet level 0 (B960 in the bench): 1024 tiles of 16 rows
    (8, 32, 32, 256, 256, 32, 32),     # Generic_UNet decoder level 0 on cat[skip, up]
    (4, 64, 64, 256, 256, 64, 8),      # flow decoder conv1 on cat[skip, up] (B64 in the bench): 1024 tiles of 8 rows
    (4, 81, 0, 256, 256, 64, 8),       # cost-volume encoder at level 0: Cin = 81
]


@pytest.mark.parametrize("B,C1,C2,H,W,Cout,groups", STREAM_CASES)
def test_conv_stream_matches_conv_f16s(dev, B, C1, C2, H, W, Cout, groups):
    """The persistent row-sharing kernel (csrc/conv_stream.hip) against the one-tile-per-workgroup kernel on the same packed weights: the same
    3-term products into one fp32 accumulator, taps summed in (kx, ky) instead of (ky, kx) order -> equal to fp32 summation-order noise
    (<= 4e-6 of the output scale), deterministic run to run; fused GroupNorm statistics likewise; one sample against an fp64 convolution."""
    from cineflow import ops
    from cineflow._lib import lib
    x1 = randn(B, C1, H, W, seed=80).to(dev)
    x2 = randn(B, C2, H, W, seed=81).to(dev) if C2 else None
    w = randn(Cout, C1 + C2, 3, 3, seed=82) / math.sqrt((C1 + C2) * 9)
    b = randn(Cout, seed=83).to(dev)
    wpk, ws = ops.pack_conv_weight_f16s(w.to(dev), c1=C1 if (C2 and C1 % 16) else None)
    prev = lib().cf_conv_stream_enable(0)
    try:
        ref_out, ref_st = ops.conv2d_f16s(x1, wpk, ws, b, Cout, 3, 3, 1, (1, 1), x2=x2, stats_groups=groups)
        ref_plain = ops.conv2d_f16s(x1, wpk, ws, None, Cout, 3, 3, 1, (1, 1), x2=x2)
        lib().cf_conv_stream_enable(2)
        out, st = ops.conv2d_f16s(x1, wpk, ws, b, Cout, 3, 3, 1, (1, 1), x2=x2, stats_groups=groups)
        plain = ops.conv2d_f16s(x1, wpk, ws, None, Cout, 3, 3, 1, (1, 1), x2=x2)
        again, st2 = ops.conv2d_f16s(x1, wpk, ws, b, Cout, 3, 3, 1, (1, 1), x2=x2, stats_groups=groups)
    finally:
        lib().cf_conv_stream_enable(prev)
    tol = 4e-6 * (1.0 + float(ref_out.abs().max()))
    assert float((out - ref_out).abs().max()) <= tol and float((plain - ref_plain).abs().max()) <= tol and torch.equal(again, out)
    scale = ref_out.double().abs().view(B, groups, -1).sum(-1)[..., None] + 1.0
    for s_ in (st, st2):
        assert float(((s_.view(B, groups, 2) - ref_st.view(B, groups, 2)).abs() / scale).max()) <= 2e-6
    xin = x1[-1:].cpu() if x2 is None else torch.cat([x1[-1:].cpu(), x2[-1:].cpu()], 1)
    ref = F.conv2d(xin.double(), w.double(), b.cpu().double(), padding=1)
    check(out[-1:], ref, 1e-5, "conv_stream vs fp64")


@pytest.mark.parametrize("C1,C2", [(81, 0), (20, 28)])
def test_conv_stream_channel_tail_never_reads_the_next_sample(dev, C1, C2):
    """the NaN-isolation property of conv_f16s' channel-tail parking, for the persistent kernel (route level 2): a batch whose odd
    samples are all NaN / Inf leaves the even samples bit-identical to a clean batch (C1 = 81: 15 zero-weight tail channels in the last chunk;
    cat[20, 28]: a tail behind each input)."""
    from cineflow import ops
    from cineflow._lib import lib
    B, H, W, Cout = 16, 128, 128, 64          # 16 x 64 tiles of 8 rows = 1024: the persistent kernel engages
    x1 = randn(B, C1, H, W, seed=70)
    x2 = randn(B, C2, H, W, seed=71) if C2 else None
    w = randn(Cout, C1 + C2, 3, 3, seed=72) / math.sqrt((C1 + C2) * 9)
    wpk, ws = ops.pack_conv_weight_f16s(w.to(dev), c1=C1 if (C2 and C1 % 16) else None)
    d = lambda t: None if t is None else t.to(dev)
    prev = lib().cf_conv_stream_enable(2)
    try:
        clean = ops.conv2d_f16s(d(x1), wpk, ws, None, Cout, 3, 3, 1, (1, 1), x2=d(x2))
        x1[1::2] = float("nan")
        if x2 is not None:
            x2[1::2] = float("inf")
        both = ops.conv2d_f16s(d(x1), wpk, ws, None, Cout, 3, 3, 1, (1, 1), x2=d(x2))
    finally:
        lib().cf_conv_stream_enable(prev)
    assert torch.isfinite(both[0::2]).all(), "even samples poisoned by the odd samples' values through the channel tail"
    assert torch.equal(both[0::2], clean[0::2])
    assert torch.isnan(both[1::2]).all()


@pytest.mark.parametrize("B,C,H,W,Cout,act", [(20, 64, 128, 128, 64, "gelu"), (32, 32, 128, 128, 32, "lrelu"), (12, 128, 128, 128, 64, "gelu"),
                                              (10, 96, 104, 136, 64, "lrelu"),
                                              (8, 32, 256, 256, 32, "lrelu")])     # the bench's map size (Generic_UNet level 0, second convolution)
def test_conv_stream_prenorm_matches(dev, B, C, H, W, Cout, act):
    """deferred input normalisation (GroupNorm / InstanceNorm + GELU / LeakyReLU applied while the tile is staged) in the persistent kernel: the
    coefficient table is double buffered per tile because consecutive tiles of a workgroup belong to different samples"""
    from cineflow import ops
    from cineflow._lib import lib
    groups = 8 if act == "gelu" else C
    raw = (randn(B, C, H, W, seed=84) * (1.0 + 0.2 * torch.arange(B).view(B, 1, 1, 1)) + 0.1 * torch.arange(B).view(B, 1, 1, 1)).to(dev)   # per-sample statistics differ
    w = (randn(Cout, C, 3, 3, seed=85) / math.sqrt(9 * C)).to(dev)
    b = randn(Cout, seed=86).to(dev)
    gam, bet = (1 + 0.1 * randn(C, seed=87)).to(dev), (0.1 * randn(C, seed=88)).to(dev)
    wpk, wsc = ops.pack_conv_weight_f16s(w)
    wsum = torch.stack([raw.double().sum((2, 3)), (raw.double() ** 2).sum((2, 3))], dim=2).view(B, groups, C // groups, 2).sum(2).reshape(-1).contiguous()
    coef = ops.group_norm_coef(wsum, gam, bet, groups, B, C, H * W)
    slope = -1.0 if act == "gelu" else 0.01
    og = Cout if act == "lrelu" else 8
    prev = lib().cf_conv_stream_enable(0)
    try:
        assert ops.prenorm_ok(raw, Cout)
        ref_out, ref_st = ops.conv2d_f16s_prenorm(raw, coef, slope, wpk, wsc, b, Cout, stats_groups=og)
        lib().cf_conv_stream_enable(2)
        assert ops.prenorm_ok(raw, Cout)
        out, st = ops.conv2d_f16s_prenorm(raw, coef, slope, wpk, wsc, b, Cout, stats_groups=og)
    finally:
        lib().cf_conv_stream_enable(prev)
    assert float((out - ref_out).abs().max()) <= 4e-6 * (1.0 + float(ref_out.abs().max()))
    scale = ref_out.double().abs().view(B, og, -1).sum(-1)[..., None] + 1.0
    assert float(((st.view(B, og, 2) - ref_st.view(B, og, 2)).abs() / scale).max()) <= 2e-6
    i = B - 1
    xn = F.group_norm(raw[i:i + 1].cpu().double(), groups, gam.cpu().double(), bet.cpu().double(), 1e-5)
    xn = F.gelu(xn) if act == "gelu" else F.leaky_relu(xn, 0.01)
    check(out[i:i + 1], F.conv2d(xn, w.cpu().double(), b.cpu().double(), padding=1), 2e-5, "conv_stream prenorm vs fp64")

