#!/usr/bin/env python3
"""Run ONE kernel a few times (for rocprofv3 --pmc runs).  python tools/one_kernel.py conv 64 0 256 64 3 1 16"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cardiac-segmentation-optical-flow_amd"))
import torch  # noqa: E402
from cineflow import ops  # noqa: E402

dev = torch.device("cuda:0")
kind = sys.argv[1]
g = torch.Generator().manual_seed(0)
if kind == "conv":
    C1, C2, H, Cout, k, stride, B = (int(v) for v in sys.argv[2:9])
    x1 = torch.randn(B, C1, H, H, generator=g).to(dev)
    x2 = torch.randn(B, C2, H, H, generator=g).to(dev) if C2 else None
    w = (torch.randn(Cout, C1 + C2, k, k, generator=g) / math.sqrt((C1 + C2) * k * k)).to(dev)
    wpk, ws = ops.pack_conv_weight_f16s(w)
    for _ in range(5):
        ops.conv2d_f16s(x1, wpk, ws, None, Cout, k, k, stride, (k // 2, k // 2), x2=x2)
elif kind == "wino":         # python tools/one_kernel.py wino C1 C2 H Cout B ntw [prenorm: gelu|lrelu]
    from cineflow._lib import lib
    C1, C2, H, Cout, B, ntw = (int(v) for v in sys.argv[2:8])
    act = sys.argv[8] if len(sys.argv) > 8 else None
    lib().cf_conv_wino_enable(ntw)
    x1 = torch.randn(B, C1, H, H, generator=g).to(dev)
    x2 = torch.randn(B, C2, H, H, generator=g).to(dev) if C2 else None
    w = (torch.randn(Cout, C1 + C2, 3, 3, generator=g) / math.sqrt((C1 + C2) * 9)).to(dev)
    wpk, wsc = ops.pack_conv_weight_wino(w)
    groups = 8 if act != "lrelu" else Cout
    if act:
        wpd, wsd = ops.pack_conv_weight_f16s(w)
        _, ws = ops.conv2d_f16s(x1, wpd, wsd, None, Cout, 3, 3, 1, (1, 1), stats_groups=groups)
        coef = ops.group_norm_coef(ws.clone(), None, None, groups, B, C1, H * H)
    for _ in range(5):
        if act:
            ops.conv2d_wino_prenorm(x1, coef, -1.0 if act == "gelu" else 0.01, wpk, wsc, None, Cout, stats_groups=groups)
        else:
            ops.conv2d_wino(x1, wpk, wsc, None, Cout, x2=x2, stats_groups=groups)
elif kind == "corr":
    C, H, s, B = (int(v) for v in sys.argv[2:6])
    a, b = torch.randn(B, C, H, H, generator=g).to(dev), torch.randn(B, C, H, H, generator=g).to(dev)
    for _ in range(5):
        ops.corr_volume(a, b, 4, s)
elif kind == "gn":
    C, H, groups, B = (int(v) for v in sys.argv[2:6])
    x = torch.randn(B, C, H, H, generator=g).to(dev)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    for _ in range(5):
        ops.group_norm(x, gam, bet, groups, act="gelu", inplace=False) if False else ops.group_norm(x, gam, bet, groups, act="gelu", out=torch.empty_like(x))
elif kind == "prenorm":      # python tools/one_kernel.py prenorm C H B gelu|lrelu
    C, H, B = (int(v) for v in sys.argv[2:5])
    act = sys.argv[5] if len(sys.argv) > 5 else "gelu"
    groups = 8 if act == "gelu" else C
    x = torch.randn(B, C, H, H, generator=g).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C)).to(dev)
    wpk, wsc = ops.pack_conv_weight_f16s(w)
    _, ws = ops.conv2d_f16s(x, wpk, wsc, None, C, 3, 3, 1, (1, 1), stats_groups=groups)
    coef = ops.group_norm_coef(ws.clone(), None, None, groups, B, C, H * H)
    for _ in range(5):
        ops.conv2d_f16s_prenorm(x, coef, -1.0 if act == "gelu" else 0.01, wpk, wsc, None, C, stats_groups=groups)
elif kind == "allpairs":     # python tools/one_kernel.py allpairs B   (fmaps [B,256,32,32], 4 levels)
    B = int(sys.argv[2])
    f1, f2 = torch.randn(B, 256, 32, 32, generator=g).to(dev), torch.randn(B, 256, 32, 32, generator=g).to(dev)
    for _ in range(5):
        ops.corr_pyramid(f1, f2, 4)
elif kind == "lookup":       # python tools/one_kernel.py lookup B
    B = int(sys.argv[2])
    f1, f2 = torch.randn(B, 256, 32, 32, generator=g).to(dev), torch.randn(B, 256, 32, 32, generator=g).to(dev)
    pyr = ops.corr_pyramid(f1, f2, 4)
    coords = ops.coords_grid(B, 32, 32, dev) + 3 * torch.randn(B, 2, 32, 32, generator=g).to(dev)
    for _ in range(5):
        ops.corr_lookup(pyr, coords, 4, 4)
elif kind == "warp":         # python tools/one_kernel.py warp B   (BASELINE config 2: warp + label warp + Jacobian of B 256x256 pairs)
    sys.path.insert(0, ROOT)
    import bench
    B = int(sys.argv[2])
    flow, img, lab = (t.to(dev) for t in bench.synthetic_pairs(B, 256, 7))
    for _ in range(3):
        ops.warp_bilinear(flow, img)
        ops.warp_labels(flow[None], lab)
        ops.jacobian_det(flow)
elif kind == "head":         # python tools/one_kernel.py head B   (norm + LeakyReLU + 1x1 head, 32 channels -> 4 classes at 256x256)
    B = int(sys.argv[2])
    raw = torch.randn(B, 32, 256, 256, generator=g).to(dev)
    ws = torch.stack([raw.double().sum((2, 3)), (raw.double() ** 2).sum((2, 3))], dim=2).reshape(-1).contiguous()
    coef = ops.group_norm_coef(ws, None, None, 32, B, 32, 65536)
    w = torch.randn(4, 32, 1, 1, generator=g).to(dev)
    for _ in range(3):
        ops.norm_head_1x1(raw, coef, 0.01, w)
elif kind == "sep":          # python tools/one_kernel.py sep 1x5|5x1 B   (SepConvGRU gates 128+256 -> 256 at 32x32)
    kh, kw = (1, 5) if sys.argv[2] == "1x5" else (5, 1)
    B = int(sys.argv[3])
    x1, x2 = torch.randn(B, 128, 32, 32, generator=g).to(dev), torch.randn(B, 256, 32, 32, generator=g).to(dev)
    w = (torch.randn(256, 384, kh, kw, generator=g) / math.sqrt(384 * 5)).to(dev)
    wpk, wsc = ops.pack_conv_weight_f16s(w)
    for _ in range(5):
        ops.conv2d_f16s(x1, wpk, wsc, None, 256, kh, kw, 1, (kh // 2, kw // 2), x2=x2)
torch.cuda.synchronize()
