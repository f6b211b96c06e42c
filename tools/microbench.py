#!/usr/bin/env python3
"""Per-kernel micro-benchmarks on the shapes of the BASELINE workload (B = slices per step).
Usage: python tools/microbench.py [--batch 16] [--only conv|gn|corr|attn|warp|pre|metrics|stem|raft|config1]"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cardiac-segmentation-optical-flow_amd"))
import torch  # noqa: E402
from cineflow import ops  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    B = args.batch
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)

    if args.only in ("", "conv"):
        print("== conv (algorithmic TFLOP/s): f16s split vs exact fp32 MFMA")
        cases = [  # C1, C2, H, Cout, k, stride
            (6, 0, 256, 64, 3, 1), (64, 0, 256, 64, 3, 1), (64, 0, 256, 128, 3, 2), (128, 0, 128, 128, 3, 1), (128, 0, 128, 256, 3, 2),
            (256, 0, 64, 256, 3, 1), (256, 0, 64, 256, 3, 2), (256, 0, 32, 256, 3, 1), (256, 256, 32, 512, 3, 1), (256, 256, 32, 256, 3, 1),
            (64, 64, 256, 64, 3, 1), (128, 128, 128, 128, 3, 1), (64, 0, 256, 2, 3, 1), (81, 0, 256, 64, 3, 1),
            (256, 0, 32, 768, 1, 1), (256, 0, 32, 2048, 1, 1), (2048, 0, 32, 256, 1, 1), (64, 0, 256, 64, 1, 1),
            (32, 0, 256, 32, 3, 1), (480, 0, 16, 480, 3, 1), (480, 0, 8, 480, 3, 1),
        ]
        for (C1, C2, H, Cout, k, stride) in cases:
            Bc = B * 4 if H <= 16 else B
            x1 = torch.randn(Bc, C1, H, H, generator=g).to(dev)
            x2 = torch.randn(Bc, C2, H, H, generator=g).to(dev) if C2 else None
            w = (torch.randn(Cout, C1 + C2, k, k, generator=g) / math.sqrt((C1 + C2) * k * k)).to(dev)
            wt = ops.prep_conv_weight(w)
            wpk, ws = ops.pack_conv_weight_f16s(w)
            pad = (k // 2, k // 2)
            Ho = (H + 2 * pad[0] - k) // stride + 1
            out = torch.empty(Bc, Cout, Ho, Ho, device=dev)
            flops = 2.0 * Bc * Ho * Ho * Cout * (C1 + C2) * k * k
            t16 = timeit(lambda: ops.conv2d_f16s(x1, wpk, ws, None, Cout, k, k, stride, pad, x2=x2, out=out))
            t32 = timeit(lambda: ops.conv2d(x1, wt, None, Cout, k, k, stride, pad, x2=x2, out=out), iters=3, warm=1)
            byts = 4.0 * (x1.numel() + (x2.numel() if C2 else 0) + out.numel())
            print("  B%3d C%4d+%-3d %3dx%-3d -> %4d k%d s%d | f16s %8.1f us %7.1f TF (%5.0f GB/s) | fp32 %9.1f us %6.1f TF" %
                  (Bc, C1, C2, H, H, Cout, k, stride, t16 * 1e6, flops / t16 / 1e12, byts / t16 / 1e9, t32 * 1e6, flops / t32 / 1e12))

    if args.only in ("", "gn"):
        print("== group norm (+GELU), bytes = read x (stats) + read x + write y")
        for (C, H, groups) in [(64, 256, 8), (128, 128, 8), (256, 64, 8), (256, 32, 8), (32, 256, 32), (480, 8, 480)]:
            x = torch.randn(B, C, H, H, generator=g).to(dev)
            gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            y = torch.empty_like(x)
            t = timeit(lambda: ops.group_norm(x, gam, bet, groups, act="gelu", out=y))
            print("  B%3d C%4d %3dx%-3d groups %3d: %8.1f us  %6.0f GB/s" % (B, C, H, H, groups, t * 1e6, 12.0 * x.numel() / t / 1e9))

    if args.only in ("", "corr"):
        print("== CorrVolume r=4 (algorithmic bytes: read cur+prev once, write 81 channels)")
        for (C, H, s) in [(64, 256, 4), (128, 128, 2), (256, 64, 1)]:
            a, b = torch.randn(B, C, H, H, generator=g).to(dev), torch.randn(B, C, H, H, generator=g).to(dev)
            t = timeit(lambda: ops.corr_volume(a, b, 4, s))
            byts = 4.0 * B * H * H * (2 * C + 81)
            fl = 2.0 * 81 * C * H * H * B
            print("  B%3d C%4d %3dx%-3d stride %d: %8.1f us  %6.0f GB/s  %6.2f TFLOP/s" % (B, C, H, H, s, t * 1e6, byts / t / 1e9, fl / t / 1e12))

    if args.only in ("", "attn"):
        print("== attention (N=1024, 4 heads x 64)")
        q = torch.randn(B, 256, 1024, generator=g).to(dev)
        t = timeit(lambda: ops.attention_cf(q, q, q, 4))
        print("  B%3d: %8.1f us  %6.2f TFLOP/s" % (B, t * 1e6, 4.0 * B * 4 * 1024 * 1024 * 64 / t / 1e12))
        x = torch.randn(B, 256, 1024, generator=g).to(dev)
        gam, bet = torch.ones(256, device=dev), torch.zeros(256, device=dev)
        t = timeit(lambda: ops.layer_norm_cf(x, gam, bet))
        print("  layer_norm_cf B%3d: %8.1f us  %6.0f GB/s" % (B, t * 1e6, 8.0 * x.numel() / t / 1e9))

    if args.only in ("", "warp"):
        print("== warp family (algorithmic bytes)")
        flow = (3 * torch.randn(B, 2, 256, 256, generator=g)).to(dev)
        for C in (1, 4):
            src = torch.randn(B, C, 256, 256, generator=g).to(dev)
            t = timeit(lambda: ops.warp_bilinear(flow, src))
            print("  warp C=%d B%3d: %8.1f us  %6.0f GB/s" % (C, B, t * 1e6, 4.0 * B * 65536 * (2 + 2 * C) / t / 1e9))
        lab = (torch.rand(B, 256, 256, generator=g) * 4).to(torch.uint8).to(dev)
        fl = flow[None].contiguous()
        t = timeit(lambda: ops.warp_labels(fl, lab))
        print("  warp_labels B%3d: %8.1f us  %6.0f GB/s" % (B, t * 1e6, B * 65536 * (8 + 1 + 1) / t / 1e9))
        t = timeit(lambda: ops.jacobian_det(flow))
        print("  jacobian B%3d: %8.1f us  %6.0f GB/s" % (B, t * 1e6, B * 65536 * (8 + 8) / t / 1e9))

    if args.only in ("", "config1"):
        # BASELINE config 1 at its stated size: one ACDC patient [1, 10, 256, 216] through the sliding-window / 4-flip TTA driver with the
        # 32-base / 6-pool U-Net of the plan (last pooling (2, 1)): host numpy volume in, host label map + softmax out (PCIe included)
        import time
        import numpy as np
        from cineflow.models import Generic_UNet
        from cineflow.inference import predict_3D_2Dconv_tiled
        from cineflow.weights import seeded_state_dict
        net = Generic_UNet(1, 32, 4, 6, pool_op_kernel_sizes=[[2, 2]] * 5 + [[2, 1]])
        net.load_state_dict(seeded_state_dict(net.state_shapes(), 41), dev)
        vol = np.random.default_rng(1).normal(size=(1, 10, 256, 216)).astype(np.float32)
        for tag, kw in (("host in / host out", {}), ("device resident outputs", {"return_device": True})):
            predict_3D_2Dconv_tiled(net, vol, (256, 224), step_size=0.5, do_mirroring=True, mirror_axes=(0, 1), use_gaussian=True, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                predict_3D_2Dconv_tiled(net, vol, (256, 224), step_size=0.5, do_mirroring=True, mirror_axes=(0, 1), use_gaussian=True, **kw)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / 5
            print("== BASELINE config 1, volume [1,10,256,216], patch (256,224), 4 flips, Generic_UNet(32, 6 pools): %7.2f ms per patient = %6.1f slices/s (%s)"
                  % (t * 1e3, 10 / t, tag))

    if args.only in ("", "pre"):
        # test-time preprocessing of one ACDC-sized case (BASELINE config 1 volume): wall time per stage, device resident, next to the
        # CPU oracle (scipy) on the same array; bytes = fp32 volume in + out
        import time
        import numpy as np
        from cineflow import preprocessing as P
        sys.path.insert(0, ROOT)
        from oracle import preprocess as OP
        print("== preprocessing, volume [1, 10, 256, 216] spacing (10, 1.5625, 1.5625) -> (10, 1.25, 1.25)")
        rng = np.random.default_rng(0)
        vol = (rng.normal(size=(1, 10, 256, 216)) * 120 + 400).astype(np.float32)
        vol[:, :, :20] = 0
        vol[:, :, :, -16:] = 0
        vd = torch.from_numpy(vol).to(dev)
        osp, tsp = np.array([10.0, 1.5625, 1.5625]), np.array([10.0, 1.25, 1.25])

        def wall(fn, n=5):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n
        t = wall(lambda: P.crop_to_nonzero(vd))
        c, sg, _ = P.crop_to_nonzero(vd)
        t0 = time.perf_counter(); OP.crop_to_nonzero(vol.copy()); tc = time.perf_counter() - t0
        print("  crop_to_nonzero          : %8.1f us   (CPU oracle %8.1f us)" % (t * 1e6, tc * 1e6))
        t = wall(lambda: P.resample_patient(c, None, osp, tsp, 3, 1, order_z_data=0, order_z_seg=0))
        r = P.resample_patient(c, None, osp, tsp, 3, 1, order_z_data=0, order_z_seg=0)[0]
        cn = c.cpu().numpy()
        t0 = time.perf_counter(); OP.resample_patient(cn.copy(), None, osp, tsp, 3, 1, order_z_data=0, order_z_seg=0); tc = time.perf_counter() - t0
        print("  resample order 3 %s -> %s: %8.1f us  %6.1f GB/s (CPU oracle %8.1f us)" % (tuple(c.shape[1:]), tuple(r.shape[1:]), t * 1e6,
                                                                                       4.0 * (c.numel() + r.numel()) / t / 1e9, tc * 1e6))
        pre = P.PreprocessorFor2D({0: "nonCT"}, {0: False}, [0, 1, 2])
        t = wall(lambda: pre.resample_and_normalize(c.clone(), tsp, {"original_spacing": osp}, sg.clone()))
        print("  resample_and_normalize   : %8.1f us" % (t * 1e6))
    if args.only in ("", "stem"):
        print("== stem convolutions (direct fp32 kernel vs the f16-split MFMA kernel), bytes = 4 B H W (Cin + Cout), fused statistics")
        for (Bs, Cin, Cout, K, groups) in [(120, 1, 32, 3, 32), (32, 1, 64, 3, 8), (32, 6, 64, 3, 8), (32, 6, 64, 1, 8)]:
            x = torch.randn(Bs, Cin, 256, 256, generator=g).to(dev)
            w = (torch.randn(Cout, Cin, K, K, generator=g) / math.sqrt(Cin * K * K)).to(dev)
            b = torch.randn(Cout, generator=g).to(dev)
            wpk, ws_ = ops.pack_conv_weight_f16s(w)
            t1 = timeit(lambda: ops.conv2d_small_cin(x, w, b, groups))
            t2 = timeit(lambda: ops.conv2d_f16s(x, wpk, ws_, b, Cout, K, K, 1, (K // 2, K // 2), stats_groups=groups))
            by = 4.0 * Bs * 65536 * (Cin + Cout)
            print("  B%3d C%d -> %2d k%d: direct %8.1f us %6.0f GB/s | f16s %8.1f us %6.0f GB/s" % (Bs, Cin, Cout, K, t1 * 1e6, by / t1 / 1e9, t2 * 1e6,
                                                                                               by / t2 / 1e9))

    if args.only in ("", "raft"):
        # BASELINE config 3: the kernels of one RAFT frame pair at 256x256 (feature maps [B,256,32,32]), B = 4 x --batch pairs
        Br = 4 * B
        print("== RAFT (config 3), B = %d frame pairs, fmaps [B,256,32,32]" % Br)
        f1, f2 = torch.randn(Br, 256, 32, 32, generator=g).to(dev), torch.randn(Br, 256, 32, 32, generator=g).to(dev)
        t = timeit(lambda: ops.corr_pyramid(f1, f2, 4))
        by = Br * 4.0 * (2 * 256 * 1024 + 1024 * 1024 * (1 + 1 / 4 + 1 / 16 + 1 / 64))
        print("  all-pairs volume + 4-level pyramid: %8.1f us  %6.0f GB/s algorithmic (7.67 MB / pair)  %6.1f TFLOP/s" %
              (t * 1e6, by / t / 1e9, 2.0 * Br * 1024 * 1024 * 256 / t / 1e12))
        pyr = ops.corr_pyramid(f1, f2, 4)
        coords = ops.coords_grid(Br, 32, 32, dev) + 3 * torch.randn(Br, 2, 32, 32, generator=g).to(dev)
        t = timeit(lambda: ops.corr_lookup(pyr, coords, 4, 4))
        print("  correlation lookup (324 ch): %8.1f us  %6.0f GB/s algorithmic (1.33 MB written / pair)" % (t * 1e6, Br * 4.0 * (324 + 2) * 1024 / t / 1e9))
        flow, mask = torch.randn(Br, 2, 32, 32, generator=g).to(dev), torch.randn(Br, 576, 32, 32, generator=g).to(dev)
        t = timeit(lambda: ops.convex_upsample(flow, mask))
        print("  convex upsample: %8.1f us  %6.0f GB/s algorithmic (2.36 MB mask + 0.52 MB out / pair)" % (t * 1e6, Br * 4.0 * 1024 * (576 + 2 + 128) / t / 1e9))
        print("  update-block convolutions at 32x32 (algorithmic TFLOP/s): f16s split vs exact fp32 MFMA")
        for (name, C1, C2, Cout, kh, kw) in [("convc1 1x1", 324, 0, 256, 1, 1), ("convc2 3x3", 256, 0, 192, 3, 3), ("convf1 7x7", 2, 0, 128, 7, 7),
                                            ("convf2 3x3", 128, 0, 64, 3, 3), ("conv 3x3", 192, 64, 126, 3, 3), ("gru rz 1x5", 128, 256, 256, 1, 5),
                                            ("gru q 1x5", 128, 256, 128, 1, 5), ("gru rz 5x1", 128, 256, 256, 5, 1), ("gru q 5x1", 128, 256, 128, 5, 1),
                                            ("flow head 3x3", 128, 0, 256, 3, 3), ("flow head 256->2", 256, 0, 2, 3, 3), ("mask 3x3", 128, 0, 256, 3, 3),
                                            ("mask 1x1", 256, 0, 576, 1, 1)]:
            x1 = torch.randn(Br, C1, 32, 32, generator=g).to(dev)
            x2 = torch.randn(Br, C2, 32, 32, generator=g).to(dev) if C2 else None
            w = (torch.randn(Cout, C1 + C2, kh, kw, generator=g) / math.sqrt((C1 + C2) * kh * kw)).to(dev)
            pad = (kh // 2, kw // 2)
            out = torch.empty(Br, Cout, 32, 32, device=dev)
            fl = 2.0 * Br * 1024 * Cout * (C1 + C2) * kh * kw
            wt = ops.prep_conv_weight(w)
            t32 = timeit(lambda: ops.conv2d(x1, wt, None, Cout, kh, kw, 1, pad, x2=x2, out=out), iters=3, warm=1)
            if ops.f16s_supported(kh, kw, 1, pad):
                wpk, wsc = ops.pack_conv_weight_f16s(w, c1=C1 if C2 else None)
                t16 = timeit(lambda: ops.conv2d_f16s(x1, wpk, wsc, None, Cout, kh, kw, 1, pad, x2=x2, out=out))
                print("    %-18s %3d+%-3d -> %3d | f16s %8.1f us %7.1f TF | fp32 %8.1f us %6.1f TF" % (name, C1, C2, Cout, t16 * 1e6, fl / t16 / 1e12,
                                                                                                 t32 * 1e6, fl / t32 / 1e12))
            else:
                print("    %-18s %3d+%-3d -> %3d | (fp32 kernel by design)           | fp32 %8.1f us %6.1f TF" % (name, C1, C2, Cout, t32 * 1e6, fl / t32 / 1e12))

    if args.only in ("", "metrics"):
        # the per-case work of compute_metrics.py (3 classes x Dice / HD / ASSD) on one ACDC-sized label volume, device vs CPU oracle
        import time
        import numpy as np
        from cineflow import metrics as M
        sys.path.insert(0, ROOT)
        from oracle import metrics as OM
        print("== metrics, label volumes [10, 256, 216], spacing (10, 1.5625, 1.5625)")
        zz, yy, xx = np.meshgrid(np.arange(10), np.arange(256), np.arange(216), indexing="ij")

        def labels(shift):
            lab = np.zeros((10, 256, 216), np.uint8)
            for c, r in ((1, 60), (2, 42), (3, 30)):
                lab[(yy - 128 - shift) ** 2 + (xx - 108 + shift) ** 2 <= (r - 2 * np.abs(zz - 5)) ** 2] = c
            return lab
        a, b, sp = labels(0), labels(3), (10.0, 1.5625, 1.5625)
        ad, bd = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)

        def case(mod, x, y):
            out = []
            for c in (1, 2, 3):
                out += [mod.dice(x == c, y == c), mod.hausdorff_distance(x == c, y == c, voxel_spacing=sp),
                        mod.avg_surface_distance_symmetric(x == c, y == c, voxel_spacing=sp)]
            return out
        case(M, ad, bd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            r = case(M, ad, bd)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 5
        t0 = time.perf_counter(); ro = case(OM, a, b); tc = time.perf_counter() - t0
        print("  Dice + HD + ASSD x 3 classes: %8.1f us   (CPU oracle %8.1f us)   max|diff| %.2e" % (t * 1e6, tc * 1e6, float(np.abs(np.array(r) - np.array(ro)).max())))
        t0 = time.perf_counter()
        for _ in range(5):
            M.label_confusion(ad, bd, 4)
        torch.cuda.synchronize()
        print("  label_confusion (all classes, one pass): %8.1f us" % ((time.perf_counter() - t0) / 5 * 1e6))

if __name__ == "__main__":
    main()
