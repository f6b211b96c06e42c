"""Per-shape time breakdown of one bench step: wraps the cineflow.ops entry points with event pairs (every call is
synchronised, so the total is slower than bench.py -- only the shares matter).  Usage: python tools/layer_profile.py [--slices 32]"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cardiac-segmentation-optical-flow_amd"))
import bench  # noqa: E402
from cineflow import ops  # noqa: E402

STATS = collections.defaultdict(lambda: [0, 0.0, 0.0])  # calls, ms, gflop


def shape_key(name, args, kw):
    if name in ("conv2d_f16s", "conv2d"):
        x1 = args[0]
        cout, kh, kw_ = args[4], args[5], args[6]
        x2 = kw.get("x2")
        c2 = 0 if x2 is None else x2.shape[1]
        st = args[7] if len(args) > 7 else kw.get("stride", 1)
        B, C1, H, W = x1.shape
        flop = 2.0 * B * (H // st) * (W // st) * cout * (C1 + c2) * kh * kw_
        return "%s B%d C%d+%d %dx%d -> %d k%d s%d%s%s" % (name, B, C1, c2, H, W, cout, kh, st, " act=" + str(kw.get("act")) if kw.get("act") else "",
                                                          " gn" if kw.get("stats_groups") else ""), flop
    if name == "conv2d_f16s_prenorm":       # (x, coef, slope, wpk, wscale, bias, cout, stats_groups=)
        x, slope, cout = args[0], args[2], args[6]
        B, C, H, W = x.shape
        return "conv2d_f16s_prenorm B%d C%d %dx%d -> %d k3 s1 %s gn" % (B, C, H, W, cout, "gelu" if slope < 0 else "lrelu"), 2.0 * B * H * W * cout * C * 9
    if name == "conv2d_wino":               # (x1, wpk, wscale, bias, cout, x2=, act=, ..., stats_groups=)
        x1, cout = args[0], args[4]
        x2 = kw.get("x2")
        c2 = 0 if x2 is None else x2.shape[1]
        B, C1, H, W = x1.shape
        return "conv2d_wino B%d C%d+%d %dx%d -> %d k3 s1%s%s" % (B, C1, c2, H, W, cout, " act=" + str(kw.get("act")) if kw.get("act") else "",
                                                                " gn" if kw.get("stats_groups") else ""), 2.0 * B * H * W * cout * (C1 + c2) * 9
    if name == "conv2d_wino_prenorm":       # (x, coef, slope, wpk, wscale, bias, cout, stats_groups=)
        x, slope, cout = args[0], args[2], args[6]
        B, C, H, W = x.shape
        return "conv2d_wino_prenorm B%d C%d %dx%d -> %d k3 s1 %s gn" % (B, C, H, W, cout, "gelu" if slope < 0 else "lrelu"), 2.0 * B * H * W * cout * C * 9
    if name == "conv2d_small_cin":          # (x, weight, bias, stats_groups)
        x, w = args[0], args[1]
        B, C, H, W = x.shape
        return "conv2d_small_cin B%d C%d %dx%d -> %d k%d" % (B, C, H, W, w.shape[0], w.shape[2]), 2.0 * B * H * W * w.shape[0] * C * w.shape[2] * w.shape[3]
    if name.startswith("conv_transpose"):
        x = args[0]
        B, C, H, W = x.shape
        cout = args[4]
        return "%s B%d C%d %dx%d -> %d" % (name, B, C, H, W, cout), 2.0 * B * H * W * 4 * C * cout
    t = [a for a in args if torch.is_tensor(a)]
    return "%s %s" % (name, "x".join(str(s) for s in t[0].shape) if t else ""), 0.0


def wrap(name):
    fn = getattr(ops, name)

    def inner(*args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args, **kw)
        e1.record()
        e1.synchronize()
        key, flop = shape_key(name, args, kw)
        s = STATS[key]
        s[0] += 1
        s[1] += e0.elapsed_time(e1)
        s[2] += flop * 1e-9
        return out

    setattr(ops, name, inner)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slices", type=int, default=32)
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--top", type=int, default=60)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    fnet, snet = bench.load_nets(bench.make_nets("video"), dev, 1234, 1, 0)
    frames = bench.synthetic_cine(a.slices, a.frames, 256, 1234).to(dev)
    lab = torch.zeros(a.slices, 256, 256, dtype=torch.uint8, device=dev)
    lab[:, 100:156, 100:156] = 1
    bench.run_step(fnet, snet, frames, lab)
    torch.cuda.synchronize()
    for n in ("conv2d_f16s", "conv2d_f16s_prenorm", "conv2d_wino", "conv2d_wino_prenorm", "conv2d_small_cin", "group_norm_coef", "conv_transpose2d_k2s2_f16s", "group_norm_apply", "group_norm", "layer_norm_cf", "attention_cf", "corr_volume",
              "warp_bilinear", "binary", "copy_channels", "gru_reset_mul", "gru_blend", "tta_accumulate", "warp_labels", "memory_input"):
        wrap(n)
    # modules bound `ops.X` at call time through the module attribute, so the wrappers are picked up
    bench.run_step(fnet, snet, frames, lab)
    torch.cuda.synchronize()
    tot = sum(s[1] for s in STATS.values())
    print("total wrapped time %.1f ms" % tot)
    for k, s in sorted(STATS.items(), key=lambda kv: -kv[1][1])[:a.top]:
        tf = s[2] / s[1] if s[1] > 0 else 0.0
        print("%6.2f%% %8.2f ms %5d calls %7.1f TF  %s" % (100 * s[1] / tot, s[1], s[0], tf, k))


if __name__ == "__main__":
    main()
