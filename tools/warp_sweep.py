"""Timing sweep of the warp family against plain copies of the same tensors (calibration of what this box streams)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cardiac-segmentation-optical-flow_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from cineflow import ops

dev = torch.device("cuda")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for B in (240, 960):
    flow, img, lab = (t.to(dev) for t in bench.synthetic_pairs(B, 256, 7))
    zero = torch.zeros_like(flow)
    px = B * 65536
    out = torch.empty_like(flow)
    rows = [("copy flow (8+8 B/px)", lambda: out.copy_(flow), 16), ("warp C=1 (16 B/px)", lambda: ops.warp_bilinear(flow, img), 16),
            ("warp C=1 zero flow", lambda: ops.warp_bilinear(zero, img), 16),
            ("labels (10 B/px)", lambda: ops.warp_labels(flow[None], lab), 10), ("labels zero flow", lambda: ops.warp_labels(zero[None], lab), 10),
            ("jacobian (16 B/px)", lambda: ops.jacobian_det(flow), 16)]
    for name, fn, bpp in rows:
        t = timeit(fn)
        print("B %4d  %-26s %8.1f us  %6.0f GB/s" % (B, name, t * 1e6, px * bpp / t / 1e9), flush=True)
