#!/usr/bin/env python3
"""Per-layer A/B of the deferred input normalisation: conv + producer's apply pass vs cf_conv2d_f16s_prenorm (tools only)."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cardiac-segmentation-optical-flow_amd"))
import torch  # noqa: E402
from cineflow import ops  # noqa: E402
from microbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
print("%-34s %9s %9s %9s   %s" % ("layer", "conv us", "apply us", "prenorm us", "gain"))
for (B, C, H, act, groups) in [(120, 32, 256, "lrelu", 32), (120, 64, 128, "lrelu", 64), (120, 128, 64, "lrelu", 128), (120, 256, 32, "lrelu", 256),
                               (120, 480, 16, "lrelu", 480), (32, 64, 256, "gelu", 8), (32, 128, 128, "gelu", 8), (32, 256, 64, "gelu", 8),
                               (32, 256, 32, "gelu", 8)]:
    x = torch.randn(B, C, H, H, generator=g).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C)).to(dev)
    b = torch.randn(C, generator=g).to(dev)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    wpk, wsc = ops.pack_conv_weight_f16s(w)
    _, ws = ops.conv2d_f16s(x, wpk, wsc, b, C, 3, 3, 1, (1, 1), stats_groups=groups)
    ws = ws.clone()
    y = torch.empty_like(x)
    t_conv = timeit(lambda: ops.conv2d_f16s(x, wpk, wsc, b, C, 3, 3, 1, (1, 1), stats_groups=groups))
    t_app = timeit(lambda: ops.group_norm_apply(x, gam, bet, groups, ws, act=act, out=y))
    if not ops.prenorm_ok(x, C):
        print("B%3d C%3d %3dx%-3d %-5s: not supported" % (B, C, H, H, act))
        continue
    coef = ops.group_norm_coef(ws, gam, bet, groups, B, C, H * H)
    t_pre = timeit(lambda: ops.conv2d_f16s_prenorm(x, coef, 0.01 if act == "lrelu" else -1.0, wpk, wsc, b, C, stats_groups=groups))
    print("B%3d C%3d %3dx%-3d %-5s            %9.1f %9.1f %9.1f   %+.1f us (%+.0f %%)" % (B, C, H, H, act, t_conv * 1e6, t_app * 1e6, t_pre * 1e6,
                                                                                    (t_conv + t_app - t_pre) * 1e6, 100 * (t_conv + t_app - t_pre) / (t_conv + t_app)))
