#!/usr/bin/env python3
"""Per-layer check of the power-limit model T = flops / 450 TF + bytes / 4.85 TB/s (profiles/r03_conv_stream.md section 7) against a layer profile
written by tools/layer_profile.py:  python tools/power_model.py profiles/r03_a_layer_profile.txt
flops = 2 MACs of the layer, bytes = inputs read once + output written once (fp32); rows = the convolution shapes of the profile."""
import re
import sys

R_TF, W_TBS = 450.0, 4.85
rows = []
for line in open(sys.argv[1]):
    m = re.match(r"\s*([\d.]+)%\s+([\d.]+) ms\s+(\d+) calls\s+([\d.]+) TF\s+(\S+) B(\d+) C(\d+)(?:\+(\d+))? (\d+)x(\d+) -> (\d+)(?: k(\d) s(\d))?(.*)", line)
    if not m:
        continue
    pct, ms, calls, _tf, name, B, C1, C2, H, W, Cout, k, s, rest = m.groups()
    pct, ms, calls, B, C1, C2, H, W, Cout = float(pct), float(ms), int(calls), int(B), int(C1), int(C2 or 0), int(H), int(W), int(Cout)
    k, s = (int(k) if k else 1), (int(s) if s else 1)
    if "transpose" in name:
        Ho, Wo, flops = 2 * H, 2 * W, 2.0 * B * H * W * C1 * Cout * 4
    else:
        Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
        flops = 2.0 * B * Ho * Wo * (C1 + C2) * Cout * k * k
    byts = 4.0 * B * ((C1 + C2) * H * W + Cout * Ho * Wo)
    t = ms / calls * 1e-3
    model = flops / (R_TF * 1e12) + byts / (W_TBS * 1e12)
    rows.append((pct, "%s B%d C%d+%d %dx%d -> %d k%d s%d %s" % (name, B, C1, C2, H, W, Cout, k, s, rest.strip()), t * 1e6, model * 1e6, t / model))
cov = sum(r[0] for r in rows)
floor = sum(r[0] / r[4] for r in rows)
print("%d convolution rows = %.1f %% of the step; the model's time for them = %.1f %% of the step (measured / model = %.3f)" % (len(rows), cov, floor, cov / floor))
print("%7s %9s %9s %6s  layer" % ("share", "measured", "model", "ratio"))
for r in sorted(rows, key=lambda r: -r[0]):
    print("%6.2f%% %7.0f us %7.0f us %6.2f  %s" % (r[0], r[2], r[3], r[4], r[1]))
