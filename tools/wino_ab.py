#!/usr/bin/env python3
"""A/B of the row-Winograd kernel (conv_wino.hip) against the direct f16-split kernel (conv_f16s.hip) on the bench's >= 128-channel 3x3 layers,
interleaved rounds in ONE process on random data (methodology rule 24): per case the median and the minimum over the rounds of
  direct | wino, 2 unit tiles per wave (two workgroups per CU) | wino, 4 unit tiles per wave (one wave per SIMD)
in microseconds and in direct-form TFLOP/s, each with fused statistics (the form the networks run) and, for single-input cases, with the
deferred normalisation.  Usage: python tools/wino_ab.py [rounds] [case indices...]"""
import math
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cardiac-segmentation-optical-flow_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from cineflow import ops  # noqa: E402
from cineflow._lib import lib  # noqa: E402
from microbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
# B, C1, C2, H, Cout, act (prenorm form when single-input), groups
CASES = [(64, 128, 0, 128, 128, "gelu", 8), (64, 256, 0, 64, 256, "gelu", 8), (64, 128, 128, 128, 128, None, 8), (64, 256, 256, 64, 256, None, 8),
         (960, 128, 0, 64, 128, "lrelu", 128), (960, 256, 0, 32, 256, "lrelu", 256), (960, 480, 0, 16, 480, "lrelu", 480), (960, 480, 480, 16, 480, None, 480),
         (64, 256, 0, 32, 256, "gelu", 8), (64, 256, 256, 32, 512, None, 0), (64, 81, 0, 128, 128, None, 8), (960, 128, 128, 64, 128, None, 128),
         (960, 256, 256, 32, 256, None, 256)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
if len(sys.argv) > 2:
    CASES = [c for i, c in enumerate(CASES) if str(i) in sys.argv[2:]]
print("rounds %d, iterations per round 5; TF = direct-form flops / time" % rounds)
for (B, C1, C2, H, Cout, act, groups) in CASES:
    x1 = torch.randn(B, C1, H, H, generator=g).to(dev)
    x2 = torch.randn(B, C2, H, H, generator=g).to(dev) if C2 else None
    w = (torch.randn(Cout, C1 + C2, 3, 3, generator=g) / math.sqrt((C1 + C2) * 9)).to(dev)
    wpd, wsd = ops.pack_conv_weight_f16s(w, c1=C1 if (C2 and C1 % 16) else None)
    wpw, wsw = ops.pack_conv_weight_wino(w, c1=C1 if (C2 and C1 % 16) else None)
    flops = 2.0 * B * H * H * Cout * (C1 + C2) * 9
    sg = groups or None
    pre = bool(act) and not C2
    if pre:
        _, ws = ops.conv2d_f16s(x1, wpd, wsd, None, Cout, 3, 3, 1, (1, 1), stats_groups=groups)
        coef = ops.group_norm_coef(ws.clone(), None, None, groups, B, C1, H * H)
        slope = 0.01 if act == "lrelu" else -1.0
    variants = {}
    if pre and ops.prenorm_ok(x1, Cout):
        variants["direct"] = lambda: ops.conv2d_f16s_prenorm(x1, coef, slope, wpd, wsd, None, Cout, stats_groups=sg)
    elif not pre:
        variants["direct"] = lambda: ops.conv2d_f16s(x1, wpd, wsd, None, Cout, 3, 3, 1, (1, 1), x2=x2, stats_groups=sg)
    else:
        variants["direct"] = lambda: ops.conv2d_f16s(x1, wpd, wsd, None, Cout, 3, 3, 1, (1, 1), stats_groups=sg)     # (+ an apply pass in the network)
    for ntw in (2, 8):
        lib().cf_conv_wino_enable(ntw)
        if ops.wino_ok(B, C1, C2, H, H, Cout, prenorm=pre):
            if pre:
                variants["wino%d" % ntw] = (lambda n: (lambda: (lib().cf_conv_wino_enable(n), ops.conv2d_wino_prenorm(x1, coef, slope, wpw, wsw, None, Cout, stats_groups=sg))))(ntw)
            else:
                variants["wino%d" % ntw] = (lambda n: (lambda: (lib().cf_conv_wino_enable(n), ops.conv2d_wino(x1, wpw, wsw, None, Cout, x2=x2, stats_groups=sg))))(ntw)
    times = {k: [] for k in variants}
    for _ in range(rounds):
        for k, fn in variants.items():
            times[k].append(timeit(fn, iters=5, warm=1))
    line = "B%3d C%3d+%-3d %3dx%-3d -> %3d %-8s" % (B, C1, C2, H, H, Cout, ("pre-" + act) if pre else "plain")
    for k in variants:
        med, mn = statistics.median(times[k]), min(times[k])
        line += " | %s %7.1f us (min %7.1f) %5.1f TF" % (k, med * 1e6, mn * 1e6, flops / med / 1e12)
    print(line, flush=True)
    lib().cf_conv_wino_enable(1)
    del x1, x2
    torch.cuda.empty_cache()
