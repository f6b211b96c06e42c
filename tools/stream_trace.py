#!/usr/bin/env python3
"""Where a conv_stream step spends its shader clocks.  Needs the trace build of the library (cd csrc && make trace), which stamps s_memtime
between the phases of every step in every wave:
    python tools/stream_trace.py            # the cases of tools/shape_ab.py that the persistent kernel takes
Phases per step: pre = statistics atomics + weight DMA issue + epilogue set-up | taps = the 54 MFMAs with their operand reads (and, in a tile's
first step, the previous tile's epilogue) | vmwait = s_waitcnt for the chunk staged next | stage = convert + ds_write | barrier."""
import ctypes
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cardiac-segmentation-optical-flow_amd")
os.environ["CINEFLOW_LIB"] = os.path.join(PKG, "cineflow", "libcineflow_hip_trace.so")
sys.path.insert(0, PKG)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cineflow import ops  # noqa: E402
from cineflow._lib import lib  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
CASES = [(240, 32, 0, 256, 32, "lrelu"), (240, 32, 32, 256, 32, None), (128, 64, 0, 256, 64, "gelu"), (128, 64, 64, 256, 64, None)]
NAMES = ["pre", "taps", "-", "-", "barrier"]
ABLATIONS = [int(a) for a in os.environ.get("ABLATIONS", "0").split(",")]      # CF_STREAM_ABLATE values to loop over (timing only)
if len(sys.argv) > 1:
    CASES = [c for i, c in enumerate(CASES) if str(i) in sys.argv[1:]]


def read_trace():
    buf = np.zeros(256 * 8 * 16, dtype=np.uint64)
    fn = lib().cf_conv_stream_trace_read
    fn.argtypes = [ctypes.c_void_p]
    fn.restype = ctypes.c_int
    torch.cuda.synchronize()
    assert fn(buf.ctypes.data) == 0
    return buf.reshape(256, 8, 16).astype(np.float64)


def report(tag, t, mfma_per_step=54):
    steps, tiles = t[..., 11], t[..., 12]
    total = t[..., 10]
    n_epi = 2 * np.maximum(tiles - 1, 0)          # the first two steps of every tile but the first carry half an epilogue each
    print("  %s: %d steps/wave, kernel %.0f clk/step (wave view; %d MFMA = %d clk of one wave's issue, x2 waves per SIMD)" % (
        tag, steps.mean(), (total / steps).mean(), mfma_per_step, mfma_per_step * 32))
    allp = t[..., 0:5].sum((0, 1)) / steps.sum()
    epi = t[..., 5:10].sum((0, 1)) / max(n_epi.sum(), 1)
    plain = (t[..., 0:5] - t[..., 5:10]).sum((0, 1)) / (steps - n_epi).sum()
    for nm, row in (("all steps", allp), ("plain step", plain), ("epilogue step", epi)):
        print("    %-14s" % nm + "  ".join("%s %6.0f" % (n, v) for n, v in zip(NAMES, row)) + "  | sum %6.0f" % row.sum())
    w = (t[..., 0:5].sum(-1) / steps).reshape(256, 8)
    print("    per-wave clk/step: min %.0f max %.0f; slowest/fastest workgroup total %.3f" % (w.min(), w.max(), total.max() / total.min()))


for (B, C1, C2, H, Cout, act) in CASES:
    x1 = torch.randn(B, C1, H, H, generator=g).to(dev)
    x2 = torch.randn(B, C2, H, H, generator=g).to(dev) if C2 else None
    w = (torch.randn(Cout, C1 + C2, 3, 3, generator=g) / math.sqrt((C1 + C2) * 9)).to(dev)
    if os.environ.get("ZERO", "0") != "0":      # power experiment: same instruction stream and memory traffic on all-zero operands
        x1.zero_()
        w.zero_()
        if x2 is not None:
            x2.zero_()
    wpk, wsc = ops.pack_conv_weight_f16s(w, c1=C1 if (C2 and C1 % 16) else None)
    groups = Cout if act == "lrelu" else 8
    print("B%d C%d+%d %dx%d -> %d" % (B, C1, C2, H, H, Cout))
    flops = 2.0 * B * H * H * Cout * (C1 + C2) * 9
    for ab in ABLATIONS:
        os.environ["CF_STREAM_ABLATE"] = str(ab)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for it in range(3):
            if it == 2:
                ev[0].record()
            _, ws = ops.conv2d_f16s(x1, wpk, wsc, None, Cout, 3, 3, 1, (1, 1), x2=x2, stats_groups=groups)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1])
        report("conv+stats ablate=%d  %.0f us %.1f TF" % (ab, ms * 1e3, flops / ms / 1e9), read_trace())
    os.environ["CF_STREAM_ABLATE"] = "0"
    _, ws = ops.conv2d_f16s(x1, wpk, wsc, None, Cout, 3, 3, 1, (1, 1), x2=x2, stats_groups=groups)
    if act and not C2 and ops.prenorm_ok(x1, Cout):
        coef = ops.group_norm_coef(ws.clone(), None, None, groups, B, C1, H * H)
        for _ in range(3):
            ops.conv2d_f16s_prenorm(x1, coef, 0.01 if act == "lrelu" else -1.0, wpk, wsc, None, Cout, stats_groups=groups)
        report("prenorm", read_trace())
    del x1, x2
    torch.cuda.empty_cache()
