#!/usr/bin/env python3
"""bench.py -- cine frames/s (seg + flow) at 256x256 on N MI355X, with the roofline of the dominant kernel and the
CPU baseline timed beside it.

A "step" is one pass of the hot path over one batch of synthetic cine slices (BASELINE.json config 4):
  B slices x T=30 frames x 256x256  ->  2-D U-Net segmentation of every frame (4-flip TTA)  +  SegFlowGaussian
  (video.yaml dispatch: CorrVolume cost volumes, 2 cross-attention bottlenecks, ConvGRU, flow decoder) run as the
  reference's two ED-anchored half sequences  +  ED-label propagation by the fused one-hot warp.
Inputs and weights are resident in HBM before the timed region; outputs stay in HBM.  frames/s = N*B*T*K / time.

Usage (driver contract):  python bench.py --gpus N --steps K --warmup W
For N>1 it is launched under torch.distributed.run, one rank per GPU; ranks run independent patient shards (weak
scaling, no data-path collective); rank 0 broadcasts the weights once over RCCL and prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cardiac-segmentation-optical-flow_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, fp32 in / fp32 accumulate
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense BF16/FP16 MFMA (never the 2:1-sparsity figure)


def synthetic_cine(B, T, S, seed):
    """bright annulus of radius 40+6cos(2 pi t/T), blobs, noise sigma 0.05, z-scored."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(S, dtype=torch.float32), torch.arange(S, dtype=torch.float32), indexing="ij")
    out = torch.empty(T, B, 1, S, S)
    for b in range(B):
        cy, cx = S / 2 + 6 * torch.randn(1, generator=g).item(), S / 2 + 6 * torch.randn(1, generator=g).item()
        blobs = [(S * torch.rand(1, generator=g).item(), S * torch.rand(1, generator=g).item(), 6 + 10 * torch.rand(1, generator=g).item())
                 for _ in range(6)]
        for t in range(T):
            r = torch.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
            rad = 40 + 6 * np.cos(2 * np.pi * t / T)
            img = torch.exp(-((r - rad) / 5.0) ** 2)
            for (by, bx, bs) in blobs:
                img = img + 0.5 * torch.exp(-((yy - by) ** 2 + (xx - bx) ** 2) / (2 * bs * bs))
            img = img + 0.05 * torch.randn(S, S, generator=g)
            out[t, b, 0] = (img - img.mean()) / img.std(unbiased=False)
    return out


def build_nets(dev, variant, seed, world, rank):
    from cineflow.models import SegFlowGaussian, Generic_UNet
    from cineflow.weights import seeded_state_dict
    from cineflow import parallel
    ma = variant == "raft_config"
    fnet = SegFlowGaussian(image_size=256, motion_appearance=ma, dim_feedforward=3072 if ma else 2048)
    snet = Generic_UNet(1, 32, 4, 6)
    for net, s in ((fnet, seed), (snet, seed + 1)):
        shapes = {k: v for k, v in net.state_shapes().items() if not k.endswith("grid")}
        sd = seeded_state_dict(shapes, s) if rank == 0 else None  # only rank 0 "reads the checkpoint"
        sd = parallel.broadcast_state_dict(sd, shapes, dev) if world > 1 else {k: v.to(dev) for k, v in sd.items()}
        net.load_state_dict(sd, dev)
    return fnet, snet


def run_step(fnet, snet, frames, ed_labels, seg_chunk):
    from cineflow.inference import predict_cine_slices
    return predict_cine_slices(fnet, _ChunkedSeg(snet, seg_chunk), frames, ed_labels)


class _ChunkedSeg:
    """Runs the segmentation U-Net over the T*B frame batch in chunks (bounds activation memory, same numbers)."""

    def __init__(self, net, chunk):
        self.net, self.chunk, self.num_classes = net, chunk, net.num_classes

    def __call__(self, x):
        if x.shape[0] <= self.chunk:
            return self.net(x)
        return torch.cat([self.net(x[i:i + self.chunk]) for i in range(0, x.shape[0], self.chunk)], 0)


def cpu_baseline(variant, seed, T_sample):
    """The oracle (CPU PyTorch restatement pinned to the reference, oracle/) on a bounded sample of the same workload:
    one slice, T_sample frames, same architectures and seeded weights, all host cores."""
    from oracle import models as OM
    from cineflow.weights import fill_module_
    from cineflow.inference import chunk_orders
    from oracle import ops as OO
    ma = variant == "raft_config"
    torch.set_num_threads(host_threads())
    fnet = fill_module_(OM.SegFlowGaussian(image_size=256, motion_appearance=ma, dim_feedforward=3072 if ma else 2048), seed)
    snet = fill_module_(OM.GenericUNet2D(1, 32, 4, 6), seed + 1)
    frames = synthetic_cine(1, T_sample, 256, 1234)
    with torch.no_grad():
        fnet(frames[:2])  # warm-up (thread pools, oneDNN primitives)
        t0 = time.perf_counter()
        probs = OM.mirror_and_predict_2d(snet, frames.reshape(T_sample, 1, 256, 256))
        seg = probs.argmax(1)
        flow = torch.zeros(T_sample, 1, 2, 256, 256)
        for order in chunk_orders(T_sample):
            if len(order) > 1:
                bf = fnet(frames[order])["backward_flow"]
                for j, t in enumerate(order[1:]):
                    flow[t] = bf[j]
        OO.warp_labels(flow, seg[:1, None].float())
        dt = time.perf_counter() - t0
    return {"value": T_sample / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "1 slice x %d frames of the same workload (oracle/ CPU PyTorch fp32), %.1f s" % (T_sample, dt)}


def log(msg):
    print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def host_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))  # a 1-GPU box gives this process a 16-core share


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--slices", type=int, default=32, help="cine slices per step and per GPU (B)")
    ap.add_argument("--frames", type=int, default=30, help="frames per cine slice (T)")
    ap.add_argument("--variant", default="video", choices=["video", "raft_config"])
    ap.add_argument("--seg-chunk", type=int, default=240, help="frames per segmentation U-Net call (240 x 32 ch x 256^2 fp32 stays under the 2 GiB buffer-offset limit of the f16 conv kernel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--conv-mode", default="f16s", choices=["f16s", "f32"], help="f16s: f16-MFMA hi/lo split (default); f32: exact fp32 MFMA")
    ap.add_argument("--cpu-frames", type=int, default=5)
    args = ap.parse_args()

    from cineflow import parallel, _lib
    rank, world, local_rank = parallel.init_from_env()
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (cineflow has no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    h = _lib.lib()
    from cineflow import ops as _ops
    _ops.set_conv_mode(args.conv_mode)

    B, T, S = args.slices, args.frames, 256
    fnet, snet = build_nets(dev, args.variant, 1234, world, rank)
    # patient i of the global list is seeded 1234+i and handled by rank i % world
    frames = synthetic_cine(B, T, S, 1234 + rank).to(dev)
    yy, xx = np.mgrid[:S, :S]
    rad = np.sqrt((yy - S / 2) ** 2 + (xx - S / 2) ** 2)
    lab = np.zeros((S, S), np.uint8)
    for k, r in enumerate((52, 44, 36), start=1):
        lab[rad < r] = k
    ed_labels = torch.from_numpy(lab)[None].repeat(B, 1, 1).contiguous().to(dev)

    log("weights + inputs resident (rank %d/%d, B=%d, T=%d)" % (rank, world, B, T))
    for _ in range(args.warmup):
        run_step(fnet, snet, frames, ed_labels, args.seg_chunk)
        torch.cuda.synchronize()
        log("warm-up step done")

    _lib.check(h.cf_profile_enable(60000), "cf_profile_enable")
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run_step(fnet, snet, frames, ed_labels, args.seg_chunk)
    torch.cuda.synchronize()
    parallel.barrier()
    dt = time.perf_counter() - t0
    dt = parallel.max_over_ranks(dt, dev)
    log("timed region: %d steps in %.3f s" % (args.steps, dt))

    # ---- roofline of the dominant kernel (conv_igemm, MFMA-bound) and of the correlation kernel (HBM-bound), from the
    # per-launch event pairs recorded during the timed steps
    def read(kid):
        ms, work, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_long()
        _lib.check(h.cf_profile_read(kid, ctypes.byref(ms), ctypes.byref(work), ctypes.byref(n)), "cf_profile_read")
        return ms.value, work.value, n.value

    conv = [read(k) for k in (0, 1, 2, 6)]
    corr = [read(k) for k in (3, 4, 5)]
    h.cf_profile_enable(0)
    dom = max(range(4), key=lambda i: conv[i][0])
    ms, flops, n = conv[dom]
    roofline = None
    if n:
        ach = flops / (ms * 1e-3) / 1e12
        if dom == 3:
            # the f16 hi/lo-split kernel issues 3 f16 MFMAs per algorithmic MAC: priced against the dense f16 MFMA peak,
            # its ceiling is 1/3; `mfma_issue_frac` is the fraction of the f16 MFMA peak the issued MFMAs reach
            roofline = {"bound": "mfma", "kernel": "conv_f16s_kernel (f16 MFMA, 3-term hi/lo split, fp32 accumulate)", "achieved": round(ach, 3),
                        "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_F16_PEAK_TFLOPS, 4),
                        "mfma_issue_frac": round(3 * ach / MFMA_F16_PEAK_TFLOPS, 4), "vs_fp32_mfma_peak": round(ach / MFMA_F32_PEAK_TFLOPS, 3),
                        "traffic": None, "launches": n, "avg_launch_us": round(ms * 1e3 / n, 2), "share_of_step_time": round(ms * 1e-3 / dt, 3)}
        else:
            roofline = {"bound": "mfma", "kernel": "conv_igemm_f32_kernel<%d,2>" % (1, 2, 4)[dom], "achieved": round(ach, 3),
                        "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                        "launches": n, "avg_launch_us": round(ms * 1e3 / n, 2), "share_of_step_time": round(ms * 1e-3 / dt, 3)}
    cms = sum(c[0] for c in corr)
    cbytes = sum(c[1] for c in corr)
    cn = sum(c[2] for c in corr)
    roofline_corr = None
    if cn:
        ach = cbytes / (cms * 1e-3) / 1e9
        roofline_corr = {"bound": "hbm", "kernel": "corr_volume_p7_kernel<1|2|4> (persistent)", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "launches": cn,
                         "avg_launch_us": round(cms * 1e3 / cn, 2),
                         "per_level": {"s%d" % s: {"GB/s": round(c[1] / (c[0] * 1e-3) / 1e9, 1), "avg_launch_us": round(c[0] * 1e3 / c[2], 2)}
                                       for s, c in zip((1, 2, 4), corr) if c[2]}}

    if rank == 0:
        frames_total = world * B * T * args.steps
        line = {
            "metric": "cine frames/sec (seg+flow) at 256x256",
            "value": round(frames_total / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16 MFMA hi/lo split, f32 accumulate (f32-class)" if args.conv_mode == "f16s" else "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config 4: joint seg+flow over 256x256x%d cine slices; Generic_UNet(32 base, 6 pools) 4-flip TTA on "
                                   "every frame + SegFlowGaussian(%s.yaml) two-chunk ED-anchored recurrence + fused label warp" % (T, args.variant),
                       "slices_per_step_per_gpu": B, "frames_per_slice": T, "image": "256x256", "sharding": "patients, rank = part_id"},
            "roofline": roofline, "roofline_corr": roofline_corr,
        }
        if world == 1 and not args.no_cpu_baseline:
            log("timing the CPU baseline (oracle, %d threads) ..." % host_threads())
            line["cpu_baseline"] = cpu_baseline(args.variant, 1234, args.cpu_frames)
            log("CPU baseline done")
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
