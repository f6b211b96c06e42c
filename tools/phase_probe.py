"""Per-phase clock sums of conv_f16s (timing build only: csrc/conv_f16s.hip compiled with -DCF_F16S_ABLATION_BUILD and loaded through
CINEFLOW_LIB).  Prints, per layer, the mean clocks a wave spends in prologue / MFMA steps / write_stage / barrier wait / epilogue."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cardiac-segmentation-optical-flow_amd"))
import torch
from cineflow import ops, _lib

dev = torch.device("cuda")
h = _lib.lib()
g = torch.Generator().manual_seed(1)
B = 32
cases = [(64, 0, 256, 64, 3, 1), (128, 0, 128, 128, 3, 1), (256, 0, 64, 256, 3, 1), (64, 64, 256, 64, 3, 1), (32, 0, 256, 32, 3, 1),
         (64, 0, 256, 128, 3, 2), (256, 0, 32, 768, 1, 1)]
for (C1, C2, H, Cout, k, stride) in cases:
    x1 = torch.randn(B, C1, H, H, generator=g).to(dev)
    x2 = torch.randn(B, C2, H, H, generator=g).to(dev) if C2 else None
    w = (torch.randn(Cout, C1 + C2, k, k, generator=g) / math.sqrt((C1 + C2) * k * k)).to(dev)
    wpk, ws = ops.pack_conv_weight_f16s(w)
    pad = (k // 2, k // 2)
    Ho = (H + 2 * pad[0] - k) // stride + 1
    out = torch.empty(B, Cout, Ho, Ho, device=dev)
    run = lambda: ops.conv2d_f16s(x1, wpk, ws, None, Cout, k, k, stride, pad, x2=x2, out=out, stats_groups=8)
    for _ in range(3):
        run()
    buf = (ctypes.c_ulonglong * 9)()
    h.cf_debug_f16s_phases(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    h.cf_debug_f16s_phases(buf)
    v = list(buf)
    waves = max(1, v[5])
    names = ["prologue", "mfma", "stage", "barrier", "epilogue"]
    tot = max(1, sum(v[:5]))
    flops = 2.0 * B * Ho * Ho * Cout * (C1 + C2) * k * k
    print("C%d+%d %dx%d -> %d k%d s%d: %.1f us %.1f TF | waves/launch %d | per wave clk: %s | total %.0f" % (
        C1, C2, H, H, Cout, k, stride, us, flops / us / 1e6, waves // n,
        "  ".join("%s %.0f (%.0f%%)" % (nm, v[i] / waves, 100.0 * v[i] / tot) for i, nm in enumerate(names)), tot / waves), flush=True)
    print("      epilogue split: scale+act %.0f  coords+stores %.0f  statistics %.0f" % (v[6] / waves, v[7] / waves, v[8] / waves), flush=True)
