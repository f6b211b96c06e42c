#!/usr/bin/env python3
"""Where the persistent Winograd kernel's waves spend their shader clocks (clock build only: make -C .../csrc clock;
CINEFLOW_LIB=.../libcineflow_hip_clock.so python tools/wino_phases.py [zeros]).  Per case: clocks per chunk of the MFMA wave (barrier wait,
compute, epilogue + item setup) and of the staging wave (transform + LDS write, load issue, barrier wait); 72 MFMAs = 2304 clocks per chunk."""
import ctypes
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cardiac-segmentation-optical-flow_amd"))
import torch  # noqa: E402
from cineflow import ops  # noqa: E402
from cineflow._lib import lib  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
zeros = len(sys.argv) > 1 and sys.argv[1] == "zeros"
fn = lib().cf_debug_wino_phases
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
lib().cf_conv_wino_enable(8)
for (B, C1, C2, H, Cout, pre) in [(128, 128, 0, 128, 128, None), (64, 256, 256, 64, 256, None), (128, 128, 0, 128, 128, "gelu"), (960, 128, 0, 64, 128, "lrelu")]:
    x1 = torch.randn(B, C1, H, H, generator=g).to(dev)
    x2 = torch.randn(B, C2, H, H, generator=g).to(dev) if C2 else None
    w = (torch.randn(Cout, C1 + C2, 3, 3, generator=g) / math.sqrt((C1 + C2) * 9)).to(dev)
    if zeros:
        x1.zero_(); w.zero_()
        if x2 is not None:
            x2.zero_()
    wpk, wsc = ops.pack_conv_weight_wino(w)
    groups = Cout if pre == "lrelu" else 8
    if pre:
        wsum = torch.stack([x1.double().sum((2, 3)), (x1.double() ** 2).sum((2, 3))], dim=2).view(B, groups, C1 // groups, 2).sum(2).reshape(-1).contiguous()
        coef = ops.group_norm_coef(wsum, None, None, groups, B, C1, H * H)
        run = lambda: ops.conv2d_wino_prenorm(x1, coef, -1.0 if pre == "gelu" else 0.01, wpk, wsc, None, Cout, stats_groups=groups)
    else:
        run = lambda: ops.conv2d_wino(x1, wpk, wsc, None, Cout, x2=x2, stats_groups=groups)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 12)()
    fn(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    fn(buf)
    v = list(buf)
    ch = max(v[7], 1)
    us = e0.elapsed_time(e1) * 100.0
    print("B%d C%d+%d %dx%d -> %d %s%s: %.0f us | per chunk: MFMA wave wait %.0f compute %.0f epilogue+setup %.0f | staging wave write %.0f issue %.0f wait %.0f clocks"
          % (B, C1, C2, H, H, Cout, pre or "plain", " (zeros)" if zeros else "", us, v[0] / ch, v[1] / ch, v[2] / ch, v[3] / ch, v[4] / ch, v[5] / ch), flush=True)
    items = ch / ((C1 + C2 + 15) // 16)
    print("    epilogue per item: output transform + activation %.0f | stores + sums %.0f | statistics reduce + atomics %.0f clocks" % (v[8] / items, v[9] / items, v[10] / items), flush=True)
    del x1, x2
    torch.cuda.empty_cache()
