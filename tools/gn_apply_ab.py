#!/usr/bin/env python3
"""Bandwidth of the GroupNorm apply pass on the bench's shapes (bytes = x + optional residual + y): python tools/gn_apply_ab.py.
Round 3: 5.5-6.1 TB/s on every shape (a device copy streams 5.2 TB/s on the same box) -- a streaming variant with compile-time activation,
four float4 per operand in flight and 64 KB per block measured the same (profiles/r03_gn_apply.txt) and was not kept."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cardiac-segmentation-optical-flow_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from cineflow import ops  # noqa: E402
from microbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (B, C, H, groups, act, with_res) in [(64, 64, 256, 8, "gelu", True), (64, 64, 256, 8, "gelu", False), (64, 128, 128, 8, "gelu", True), (64, 256, 64, 8, "gelu", True),
                                         (240, 32, 256, 32, "lrelu", False), (240, 64, 128, 64, "lrelu", False)]:
    x = torch.randn(B, C, H, H, generator=g).to(dev)
    r = torch.randn(B, C, H, H, generator=g).to(dev) if with_res else None
    gam, bet = (1 + 0.1 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    xs = x.double().view(B, groups, -1)
    ws = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], dim=2).reshape(-1).contiguous()      # (sum, sum of squares) per (sample, group), fp64
    del xs
    y = torch.empty_like(x)
    kw = dict(act=act, out=y)
    if with_res:
        kw.update(res=r, res_mode="after_act")
    t = timeit(lambda: ops.group_norm_apply(x, gam, bet, groups, ws, 1e-5, **kw))
    nb = (3 if with_res else 2) * x.numel() * 4
    print("  B%3d C%3d %3dx%-3d %-5s res=%d: %8.1f us  %5.2f TB/s" % (B, C, H, H, act, with_res, t * 1e6, nb / t / 1e12), flush=True)
    del x, r, y
    torch.cuda.empty_cache()
