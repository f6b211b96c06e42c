#!/usr/bin/env python3
"""Board power and shader clock (rocm-smi, polled from a thread) while one kernel runs back to back for a few seconds: is a layer's time set by the
power limit?  python tools/power_probe.py   -> one line per workload: average / peak power, average sclk, achieved TF or TB/s.
With the clock build (make -C .../csrc clock; CINEFLOW_LIB=.../libcineflow_hip_clock.so python tools/power_probe.py) every convolution line also
carries the IN-KERNEL clock: delta s_memtime / delta s_memrealtime x 100 MHz summed over the workgroups of the run (MI355X_MICROARCH.md, DVFS
give-back item 6) -- the test the guide prescribes; rocm-smi's sclk is a coarser neighbour of it."""
import math
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cardiac-segmentation-optical-flow_amd"))
import torch  # noqa: E402
from cineflow import ops  # noqa: E402

import ctypes  # noqa: E402
from cineflow._lib import lib  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
HAVE_CLOCK = hasattr(lib(), "cf_debug_clock_f16s")


def kernel_clock():
    """in-kernel clock of the convolutions launched since the last call (clock build only): 'x.xx GHz (n workgroups)' per kernel family"""
    if not HAVE_CLOCK:
        return ""
    out = []
    for name in ("f16s", "wino"):
        fn = getattr(lib(), "cf_debug_clock_" + name)
        fn.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long)]
        ghz, n = ctypes.c_double(), ctypes.c_long()
        fn(ctypes.byref(ghz), ctypes.byref(n))
        if n.value:
            out.append("in-kernel clock %s %.3f GHz (%d workgroups)" % (name, ghz.value, n.value))
    return " | " + ", ".join(out) if out else ""
samples = []
stop = False


def poll():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            pw = re.search(r"(?:Average|Current Socket) Graphics Package Power \(W\):\s*([0-9.]+)", o)
            ck = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", o)
            samples.append((time.perf_counter(), float(pw.group(1)) if pw else float("nan"), float(ck.group(1)) if ck else float("nan")))
        except Exception as e:      # noqa: BLE001
            samples.append((time.perf_counter(), float("nan"), float("nan")))
        time.sleep(0.2)


def run(tag, fn, work, unit, secs=4.0):
    fn()
    torch.cuda.synchronize()
    kernel_clock()           # reset the stamps: only the timed launches below count
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(10):
            fn()
        n += 10
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    mine = [s for s in samples if t0 + 0.7 <= s[0] <= t1]
    pw = [s[1] for s in mine if s[1] == s[1]]
    ck = [s[2] for s in mine if s[2] == s[2]]
    print("%-44s %7.1f %s | power avg %6.0f W peak %6.0f W (%d samples) | sclk avg %5.0f MHz%s" % (
        tag, work * n / (t1 - t0), unit, sum(pw) / max(len(pw), 1), max(pw) if pw else float("nan"), len(pw), sum(ck) / max(len(ck), 1), kernel_clock()), flush=True)


def conv_case(B, C, H, Cout, zero=False):
    x = torch.randn(B, C, H, H, generator=g).to(dev)
    w = (torch.randn(Cout, C, 3, 3, generator=g) / math.sqrt(C * 9)).to(dev)
    if zero:
        x.zero_()
        w.zero_()
    wpk, wsc = ops.pack_conv_weight_f16s(w)
    return (lambda: ops.conv2d_f16s(x, wpk, wsc, None, Cout, 3, 3, 1, (1, 1), stats_groups=8)), 2.0 * B * H * H * Cout * C * 9 / 1e12


def wino_case(B, C1, C2, H, Cout, zero=False):
    x1 = torch.randn(B, C1, H, H, generator=g).to(dev)
    x2 = torch.randn(B, C2, H, H, generator=g).to(dev) if C2 else None
    w = (torch.randn(Cout, C1 + C2, 3, 3, generator=g) / math.sqrt((C1 + C2) * 9)).to(dev)
    if zero:
        x1.zero_()
        w.zero_()
        if x2 is not None:
            x2.zero_()
    wpk, wsc = ops.pack_conv_weight_wino(w)
    return (lambda: ops.conv2d_wino(x1, wpk, wsc, None, Cout, x2=x2, stats_groups=8)), 2.0 * B * H * H * Cout * (C1 + C2) * 9 / 1e12


th = threading.Thread(target=poll, daemon=True)
th.start()
time.sleep(1.5)
print(subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True).stdout.strip().splitlines()[-3:])
t0 = time.perf_counter()
time.sleep(2.0)
idle = [s[1] for s in samples if s[1] == s[1]]
print("idle: %.0f W" % (sum(idle) / max(len(idle), 1)))
only_corr = len(sys.argv) > 1 and sys.argv[1] == "corr"
if len(sys.argv) > 1 and sys.argv[1] == "wino":
    # the two Winograd cases only (energy ablations: CINEFLOW_LIB = a clock build made with ABL=n, wrong results by design)
    f, w_ = wino_case(128, 128, 0, 128, 128)
    run("wino 128 -> 128, 128x128, B128 (random)", f, w_, "TF")
    f, w_ = wino_case(64, 256, 256, 64, 256)
    run("wino 256+256 -> 256, 64x64, B64 (random)", f, w_, "TF")
    stop = True
    sys.exit(0)
f, w_ = conv_case(128, 128, 128, 128)
run("conv 128 -> 128, 128x128, B128 (random)", f, w_, "TF", secs=1.5 if only_corr else 4.0)
if not only_corr:
    f, w_ = conv_case(128, 128, 128, 128, zero=True)
    run("conv 128 -> 128, 128x128, B128 (zeros)", f, w_, "TF")
    f, w_ = wino_case(128, 128, 0, 128, 128)
    run("wino 128 -> 128, 128x128, B128 (random)", f, w_, "TF")
    f, w_ = wino_case(128, 128, 0, 128, 128, zero=True)
    run("wino 128 -> 128, 128x128, B128 (zeros)", f, w_, "TF")
    f, w_ = wino_case(64, 256, 256, 64, 256)
    run("wino 256+256 -> 256, 64x64, B64 (random)", f, w_, "TF")
    f, w_ = wino_case(64, 256, 256, 64, 256, zero=True)
    run("wino 256+256 -> 256, 64x64, B64 (zeros)", f, w_, "TF")
    f, w_ = conv_case(128, 64, 256, 64)
    run("conv 64 -> 64, 256x256, B128 (random)", f, w_, "TF")
    f, w_ = conv_case(128, 64, 256, 64, zero=True)
    run("conv 64 -> 64, 256x256, B128 (zeros)", f, w_, "TF")
    x = torch.randn(128, 64, 256, 256, generator=g).to(dev)
    y = torch.empty_like(x)
    run("device copy 2.1 GB (read + write)", lambda: y.copy_(x), 2 * x.numel() * 4 / 1e12, "TB/s")
    gam, bet = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    run("group_norm + GELU apply, 128x64x256x256", lambda: ops.group_norm(x, gam, bet, 8, act="gelu", out=y), 2 * x.numel() * 4 / 1e12, "TB/s")
if len(sys.argv) > 1 and sys.argv[1] == "corr":
    for (C, H, st) in [(64, 256, 4), (128, 128, 2), (256, 64, 1)]:
        a = torch.randn(128, C, H, H, generator=g).to(dev)
        b = torch.randn(128, C, H, H, generator=g).to(dev)
        run("corr_volume C=%d %dx%d dilation %d, B=128" % (C, H, H, st), lambda: ops.corr_volume(a, b, 4, st), 4.0 * 128 * H * H * (2 * C + 81) / 1e12, "TB/s")
        a.zero_()
        b.zero_()
        run("   the same on zeros", lambda: ops.corr_volume(a, b, 4, st), 4.0 * 128 * H * H * (2 * C + 81) / 1e12, "TB/s")
        del a, b
        torch.cuda.empty_cache()
stop = True
