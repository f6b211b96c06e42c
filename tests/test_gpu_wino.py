"""GPU parity of the row-Winograd F(2,3) convolution (csrc/conv_wino.hip, cf_conv2d_wino / cf_conv2d_wino_prenorm) against an fp64
convolution of the same operands, at the layer shapes of the bench networks (batch reduced), on every kernel form (route level 2 / 4: one tile per workgroup with 2 / 4
unit tiles per wave; 8: the persistent wave-specialised kernel).

Tolerance 2e-5 of the output scale (the outputs are O(1): unit-variance inputs, He-scaled weights): the kernel carries ~2^-22 relative
operand error and fp32 accumulation like cf_conv2d_f16s (measured 3-5e-7 of max|y|, tools/winograd_eval.py --row)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def randn(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def maxdiff(a, b):
    return float((a.detach().cpu().double() - torch.as_tensor(b).double()).abs().max())


@pytest.fixture()
def wino_level():
    """restores the library's route level after a test that forces a workgroup shape"""
    from cineflow._lib import lib
    prev = lib().cf_conv_wino_enable(1)
    yield lib().cf_conv_wino_enable
    lib().cf_conv_wino_enable(prev)


# B, C1, C2, H, W, Cout, act, statistics groups
WINO_CASES = [
    (2, 128, 0, 128, 128, 128, None, 8),        # flow encoder conv, B64 in the bench
    (2, 256, 0, 64, 64, 256, None, 8),
    (2, 128, 128, 128, 128, 128, None, 8),      # decoder conv1 on cat[skip, up]
    (2, 256, 256, 64, 64, 256, None, 8),
    (3, 128, 0, 64, 64, 128, None, 128),        # U-Net stage (InstanceNorm statistics), B960 in the bench
    (3, 256, 0, 32, 32, 256, None, 256),
    (2, 480, 0, 16, 16, 480, None, 480),        # 480 = 3 x 128 + 96 output channels, 16-wide maps (8 units per row)
    (2, 480, 480, 16, 16, 480, None, 480),
    (2, 81, 0, 128, 128, 128, None, 8),         # cost-volume encoder: channel tail inside the last chunk
    (2, 256, 256, 32, 32, 512, "sigmoid", 0),   # ConvGRU gates
    (2, 256, 256, 32, 32, 256, "tanh", 0),      # ConvGRU candidate
    (1, 100, 60, 32, 64, 224, None, 8),         # split not a chunk multiple (split-aware packing), 224 = 128 + 96, ragged everything else
]


@pytest.mark.parametrize("ntw", [2, 4, 8])
@pytest.mark.parametrize("case", WINO_CASES)
def test_conv2d_wino_vs_fp64(dev, wino_level, case, ntw):
    from cineflow import ops
    B, C1, C2, H, W, Cout, act, groups = case
    wino_level(ntw)
    if not ops.wino_ok(B, C1, C2, H, W, Cout):
        pytest.skip("shape not built for route level %d" % ntw)
    x1 = randn(B, C1, H, W, seed=30)
    x2 = randn(B, C2, H, W, seed=31) if C2 else None
    w = randn(Cout, C1 + C2, 3, 3, seed=32) / math.sqrt((C1 + C2) * 9)
    b = randn(Cout, seed=33)
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    ref = F.conv2d(xin.double(), w.double(), b.double(), padding=1)
    want = {"sigmoid": torch.sigmoid, "tanh": torch.tanh, None: lambda t: t}[act](ref)
    wpk, ws = ops.pack_conv_weight_wino(w.to(dev), c1=C1 if C2 else None)
    got = ops.conv2d_wino(x1.to(dev), wpk, ws, b.to(dev), Cout, x2=None if x2 is None else x2.to(dev), act=act, stats_groups=groups or None)
    if groups:
        got, st = got
        yo = got.cpu().double().view(B, groups, -1)
        wst = torch.stack([yo.sum(-1), (yo ** 2).sum(-1)], -1)
        assert float(((st.cpu().view(B, groups, 2) - wst).abs() / (yo.abs().sum(-1)[..., None] + 1.0)).max()) <= 2e-6, "fused statistics"
    scale = float(want.abs().max())
    d = maxdiff(got, want)
    assert d <= 2e-5 * max(scale, 1.0), "conv_wino max|diff| %.3e (scale %.2f)" % (d, scale)
    # the direct kernel on the same operands: same result to fp32 summation noise
    wpd, wsd = ops.pack_conv_weight_f16s(w.to(dev), c1=C1 if C2 else None)
    direct = ops.conv2d_f16s(x1.to(dev), wpd, wsd, b.to(dev), Cout, 3, 3, 1, (1, 1), x2=None if x2 is None else x2.to(dev), act=act)
    assert maxdiff(got, direct.cpu()) <= 1e-5 * max(scale, 1.0)


@pytest.mark.parametrize("ntw", [2, 4, 8])
def test_conv2d_wino_epilogue_slice_and_residual(dev, wino_level, ntw):
    """output written into a channel slice of a wider tensor, residual add, GELU; the other channels stay untouched"""
    from cineflow import ops
    wino_level(ntw)
    B, C, H, W, Cout = 2, 64, 32, 32, 128
    x = randn(B, C, H, W, seed=1)
    w = randn(Cout, C, 3, 3, seed=2) / math.sqrt(C * 9)
    b = randn(Cout, seed=3)
    res = randn(B, Cout, H, W, seed=4)
    want = F.gelu(F.conv2d(x.double(), w.double(), b.double(), padding=1)) + res.double()
    wpk, ws = ops.pack_conv_weight_wino(w.to(dev))
    big = torch.full((B, Cout + 5, H, W), 7.0, device=dev)
    ops.conv2d_wino(x.to(dev), wpk, ws, b.to(dev), Cout, act="gelu", res=res.to(dev), out=big, out_coff=3)
    assert maxdiff(big[:, 3:3 + Cout], want) <= 2e-5
    assert float(big[:, :3].min()) == 7.0 and float(big[:, :3].max()) == 7.0 and float(big[:, 3 + Cout:].min()) == 7.0 and float(big[:, 3 + Cout:].max()) == 7.0


@pytest.mark.parametrize("ntw", [2, 4, 8])
@pytest.mark.parametrize("B,C,H,W,Cout", [(2, 128, 128, 128, 128), (2, 256, 64, 64, 256), (3, 128, 64, 64, 128), (3, 256, 32, 32, 256), (2, 480, 16, 16, 480)])
def test_conv2d_wino_prenorm(dev, wino_level, B, C, H, W, Cout, ntw):
    """deferred InstanceNorm + LeakyReLU (Generic_UNet) and GroupNorm(8) + GELU (DoubleConv) applied while the tile is staged, against torch
    on the materialised activation"""
    from cineflow import ops
    wino_level(ntw)
    if not ops.wino_ok(B, C, 0, H, W, Cout, prenorm=True):
        pytest.skip("shape not built for route level %d" % ntw)
    x = randn(B, C, H, W, seed=100) * 1.7 + 0.4
    g, bt = randn(C, seed=101), randn(C, seed=102)
    w = randn(Cout, C, 3, 3, seed=103) / math.sqrt(C * 9)
    b = randn(Cout, seed=104)
    xd = x.to(dev)
    wpk, wsc = ops.pack_conv_weight_wino(w.to(dev))
    xs = x.double().view(B, C, -1)
    ws = torch.stack([xs.sum(-1), (xs ** 2).sum(-1)], -1).reshape(-1).to(dev)
    coef = ops.group_norm_coef(ws, g.to(dev), bt.to(dev), C, B, C, H * W)
    want = F.conv2d(F.leaky_relu(F.instance_norm(x.double(), weight=g.double(), bias=bt.double(), eps=1e-5), 0.01), w.double(), b.double(), padding=1)
    out, st = ops.conv2d_wino_prenorm(xd, coef, 0.01, wpk, wsc, b.to(dev), Cout, stats_groups=Cout)
    scale = float(want.abs().max())
    assert maxdiff(out, want) <= 3e-5 * max(scale, 1.0)
    yo = out.cpu().double().view(B, Cout, -1)
    wst = torch.stack([yo.sum(-1), (yo ** 2).sum(-1)], -1)
    assert float(((st.cpu().view(B, Cout, 2) - wst).abs() / (yo.abs().sum(-1)[..., None] + 1.0)).max()) <= 2e-6
    xg = x.double().view(B, 8, -1)
    wsg = torch.stack([xg.sum(-1), (xg ** 2).sum(-1)], -1).reshape(-1).to(dev)
    coefg = ops.group_norm_coef(wsg, g.to(dev), bt.to(dev), 8, B, C, H * W)
    wantg = F.conv2d(F.gelu(F.group_norm(x.double(), 8, g.double(), bt.double(), eps=1e-5)), w.double(), b.double(), padding=1)
    assert maxdiff(ops.conv2d_wino_prenorm(xd, coefg, -1.0, wpk, wsc, b.to(dev), Cout), wantg) <= 3e-5 * max(float(wantg.abs().max()), 1.0)


@pytest.mark.parametrize("B,C1,C2,H,W,Cout,pre", [
    (72, 144, 0, 32, 32, 128, False),      # 576 items on 256 workgroups: 2-3 items each, 9 chunks per item -> odd AND even chunk streams
    (40, 64, 48, 32, 64, 256, False),      # two inputs, two 128-channel blocks: 1280 items, 7 chunks per item, bands cross samples and blocks
    (72, 128, 0, 32, 32, 128, True),       # deferred GroupNorm + GELU: the coefficient quads follow the stream across samples
    (48, 256, 0, 16, 16, 256, True),       # 16-wide maps (8 units per row), 16 chunks
])
def test_conv2d_wino_persistent_stream_across_items(dev, wino_level, B, C1, C2, H, W, Cout, pre):
    """The persistent kernel at launch sizes where a workgroup walks SEVERAL items: its staging waves run the chunk stream across item (and
    sample) boundaries two chunks of loads ahead, with one barrier per pair of chunks over four LDS buffers -- odd streams, items of different
    samples in one stream and the padded last pair are only reached when n_items > n_CUs.  Reference: the direct kernel on the same operands
    (1e-5: same products, different summation order) and an fp64 convolution on the first and last two samples (2e-5)."""
    from cineflow import ops
    wino_level(8)
    assert ops.wino_ok(B, C1, C2, H, W, Cout, prenorm=pre)
    x1 = randn(B, C1, H, W, seed=200) * 1.3 + 0.2
    x2 = randn(B, C2, H, W, seed=201) if C2 else None
    w = randn(Cout, C1 + C2, 3, 3, seed=202) / math.sqrt((C1 + C2) * 9)
    b = randn(Cout, seed=203)
    x1d, x2d = x1.to(dev), None if x2 is None else x2.to(dev)
    wpk, ws = ops.pack_conv_weight_wino(w.to(dev), c1=C1 if C2 else None)
    wpd, wsd = ops.pack_conv_weight_f16s(w.to(dev), c1=C1 if C2 else None)
    if pre:
        g, bt = randn(C1, seed=204), randn(C1, seed=205)
        xs = x1.double().view(B, 8, -1)
        wsum = torch.stack([xs.sum(-1), (xs ** 2).sum(-1)], -1).reshape(-1).to(dev)
        coef = ops.group_norm_coef(wsum, g.to(dev), bt.to(dev), 8, B, C1, H * W)
        got, st = ops.conv2d_wino_prenorm(x1d, coef, -1.0, wpk, ws, b.to(dev), Cout, stats_groups=8)
        wino_level(0)
        direct = ops.conv2d_f16s_prenorm(x1d, coef, -1.0, wpd, wsd, b.to(dev), Cout) if ops.prenorm_ok(x1d, Cout) else None
        sel = [0, 1, B - 2, B - 1]
        xin = F.gelu(F.group_norm(x1[sel].double(), 8, g.double(), bt.double(), eps=1e-5))
    else:
        got, st = ops.conv2d_wino(x1d, wpk, ws, b.to(dev), Cout, x2=x2d, stats_groups=8)
        wino_level(0)
        direct = ops.conv2d_f16s(x1d, wpd, wsd, b.to(dev), Cout, 3, 3, 1, (1, 1), x2=x2d)
        sel = [0, 1, B - 2, B - 1]
        xin = x1[sel].double() if x2 is None else torch.cat([x1[sel], x2[sel]], 1).double()
    want = F.conv2d(xin, w.double(), b.double(), padding=1)
    scale = max(float(want.abs().max()), 1.0)
    assert maxdiff(got[sel], want) <= 2e-5 * scale
    if direct is not None:
        assert maxdiff(got, direct.cpu()) <= 1e-5 * scale
    yo = got.cpu().double().view(B, 8, -1)
    wst = torch.stack([yo.sum(-1), (yo ** 2).sum(-1)], -1)
    assert float(((st.cpu().view(B, 8, 2) - wst).abs() / (yo.abs().sum(-1)[..., None] + 1.0)).max()) <= 2e-6, "fused statistics"


def test_conv2d_wino_capability_and_errors(dev, wino_level):
    from cineflow import ops
    from cineflow._lib import CineflowError
    assert ops.wino_ok(2, 128, 0, 64, 64, 128) and ops.wino_ok(2, 480, 480, 16, 16, 480) and ops.wino_ok(1, 256, 0, 32, 32, 512)
    assert not ops.wino_ok(2, 64, 0, 64, 64, 64)         # 64 output channels: the direct kernels
    assert not ops.wino_ok(2, 128, 0, 8, 8, 128)         # 8-wide maps
    assert not ops.wino_ok(2, 128, 0, 30, 40, 128)       # W % 16 != 0
    assert not ops.wino_ok(2, 128, 0, 10, 32, 128)       # H not a multiple of the tile rows
    assert not ops.wino_ok(2, 128, 128, 64, 64, 128, prenorm=True)    # deferred normalisation: single input only
    wino_level(0)
    assert not ops.wino_ok(2, 128, 0, 64, 64, 128)
    wino_level(1)
    w = randn(128, 128, 3, 3, seed=1).to(dev)
    wpk, ws = ops.pack_conv_weight_wino(w)
    with pytest.raises(CineflowError):
        ops.conv2d_wino(torch.zeros(2, 128, 30, 40, device=dev), wpk, ws, None, 128)


def test_conv2d_wino_nonfinite_and_channel_tail(dev, wino_level):
    """NaN / Inf inputs propagate as through an fp32 convolution and stay inside their sample; the zero-weight channel tail of the last chunk
    never reads the next sample's first channels (C1 = 81: 15 padded channels)"""
    from cineflow import ops
    B, C, H, W, Cout = 2, 81, 32, 32, 128
    x = randn(B, C, H, W, seed=5).to(dev)
    w = (randn(Cout, C, 3, 3, seed=6) / math.sqrt(C * 9)).to(dev)
    wpk, ws = ops.pack_conv_weight_wino(w)
    clean = ops.conv2d_wino(x, wpk, ws, None, Cout)
    xp = x.clone()
    xp[1, 0] = float("nan")                  # the channels right behind sample 0's tail
    out = ops.conv2d_wino(xp, wpk, ws, None, Cout)
    assert torch.equal(out[0], clean[0]) and bool(torch.isnan(out[1]).all())
    xq = x.clone()
    xq[0, 3, 10, 10] = float("inf")
    out = ops.conv2d_wino(xq, wpk, ws, None, Cout)
    assert not bool(torch.isfinite(out[0, :, 10, 10]).any()) and torch.equal(out[1], clean[1])
