#!/usr/bin/env python3
"""bench.py -- cine frames/s (seg + flow) at 256x256 on N MI355X, with the roofline of the dominant kernel, the
correlation kernels' HBM rooflines and the CPU baseline timed beside it.

A "step" is one pass of the hot path over one batch of synthetic cine slices (BASELINE.json config 4):
  B slices x T=30 frames x 256x256  ->  2-D U-Net segmentation of every frame (4-flip TTA)  +  SegFlowGaussian
  (video.yaml dispatch: CorrVolume cost volumes, 2 cross-attention bottlenecks, ConvGRU, flow decoder) run as the
  reference's two ED-anchored half sequences  +  ED-label propagation by the fused one-hot warp.
Inputs and weights are resident in HBM before the timed region; outputs stay in HBM.  frames/s = N*B*T*K / time.

BASELINE config 3 (RAFT: all-pairs correlation pyramid + 12 update iterations per 256x256 frame pair,
SegFlowGaussian.py:875-969) is measured by the same command as a nested object `config3_raft` of the ONE JSON line
(`--variant raft` makes it the headline of the line instead; `--no-raft` skips it).
Other variants: `successive` = config 4 with its other dispatch (successive.yaml: OpticalFlowModelSuccessive x 2 in ModelWrap, d_model 512,
8 heads); `warp` = BASELINE config 2 (VoxelMorph warp of frame pairs); `api` = the file-level API end to end (`predict_from_folder` on
synthetic patients in a tmpfs folder: NIfTI read + preprocessing, device, NIfTI / NPZ export) with the wall-time split.

Usage (driver contract):  python bench.py --gpus N --steps K --warmup W
  * N > 1 without WORLD_SIZE in the environment: this process starts N ranks of itself (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, one per GPU) BEFORE anything touches the GPU, relays rank 0's JSON line and exits non-zero if a rank
    fails.  Under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` it is one of the ranks.
  * ranks run independent patient shards (weak scaling, no data-path collective); rank 0 "reads the checkpoint" and
    broadcasts the weights once over RCCL; the step time is the MAX over ranks; rank 0 prints ONE JSON line.
  * --dry-run: the same launch / rendezvous / broadcast / barrier / max-over-ranks path on gloo + CPU tensors with the
    GPU work replaced by a sleep (tests/test_distributed_cpu.py runs it at N = 2 in the build container).
"""
import argparse
import ctypes
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cardiac-segmentation-optical-flow_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, fp32 in / fp32 accumulate
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense BF16/FP16 MFMA (never the 2:1-sparsity figure)
# what a register-operand loop of v_mfma_f32_32x32x16_f16 sustains on this pool's MI355X with non-zero operands (the clock drops from
# 2.37 to 1.64 GHz; all-zero operands reach 2482): tools/ubench/mfma_rate.hip, profiles/r02_mfma_rate.txt.  Reported beside `peak`, never instead.
MFMA_F16_SUSTAINED_TFLOPS = 1720.0
# HBM bytes per launch over algorithmic bytes, from the separate PMC passes (FETCH_SIZE / WRITE_SIZE with the guide's wide-load correction):
# the live line carries `traffic` = algorithmic x this ratio with its source, it cannot collect counters itself (gpurun refuses --pmc next to
# tracing, and a PMC pass serialises kernels).
# Matrix-pipe busy fraction of KERNEL WALL TIME per dominant convolution shape, from one GRBM_GUI_ACTIVE + SQ_VALU_MFMA_BUSY_CYCLES pass each
# (tools/pmc_busy.sh; busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8); effective clock of the profiled pass = GRBM_GUI_ACTIVE / 8 /
# wall time) and the in-kernel clock of a plain run (delta s_memtime / delta s_memrealtime, tools/power_probe.py on the clock build).  Constants
# with their source, like PMC_TRAFFIC_RATIO: a counter pass cannot run inside the timed region.
MFMA_BUSY = {"source": "profiles/r04_power_probe.txt (tools/pmc_busy.sh passes + tools/power_probe.py on the clock build)",
             "conv_wino_ps_kernel 128 -> 128 @128x128 B128": {"mfma_busy_of_wall": 0.46, "clock_GHz_profiled_pass": 1.51, "in_kernel_clock_GHz_plain_run": 1.84,
                                                             "in_kernel_clock_GHz_zero_operands": 2.39},
             "conv_wino_ps_kernel 256+256 -> 256 @64x64 B64": {"mfma_busy_of_wall": 0.59, "clock_GHz_profiled_pass": 1.39, "in_kernel_clock_GHz_plain_run": 1.70,
                                                              "in_kernel_clock_GHz_zero_operands": 2.39},
             "conv_stream_kernel 64 -> 64 @256x256 B128": {"mfma_busy_of_wall": 0.54, "clock_GHz_profiled_pass": 1.48},
             "conv_f16s_kernel 128 -> 128 @128x128 B128 (direct form, for comparison)": {"mfma_busy_of_wall": 0.76, "clock_GHz_profiled_pass": 1.29,
                                                                                         "in_kernel_clock_GHz_plain_run": 1.53,
                                                                                         "in_kernel_clock_GHz_zero_operands": 2.38}}

PMC_TRAFFIC_RATIO = {"conv_f16s": (1.00, "profiles/r02_pmc_hbm_traffic.md, r03_pmc_hbm_traffic.md (reads 1.00-1.03x, writes 1.00x algorithmic)"),
                     "corr": (1.13, "profiles/r04_pmc_hbm_traffic.md (corr_volume_mfma_kernel: reads 1.29x at dilation 4, 1.08x at 2, 1.00x at 1, writes exact; byte-weighted 1.13x)")}


def synthetic_cine(B, T, S, seed):
    """bright annulus of radius 40+6cos(2 pi t/T), six blobs, noise sigma 0.05, z-scored per
    frame.  Vectorised over (T, B): a fraction of a second for 32 x 30 frames (the per-frame Python loop took 50 s)."""
    g = torch.Generator().manual_seed(seed)
    ax = torch.arange(S, dtype=torch.float32)
    yy, xx = ax.view(1, 1, S, 1), ax.view(1, 1, 1, S)
    c = S / 2 + 6 * torch.randn(2, B, generator=g)
    blob = torch.rand(B, 6, 3, generator=g)
    by, bx, bs = S * blob[..., 0], S * blob[..., 1], 6 + 10 * blob[..., 2]
    r = torch.sqrt((yy - c[0].view(1, B, 1, 1)) ** 2 + (xx - c[1].view(1, B, 1, 1)) ** 2)         # [1,B,S,S]
    rad = (40 + 6 * torch.cos(2 * np.pi * torch.arange(T, dtype=torch.float32) / T)).view(T, 1, 1, 1)
    img = torch.exp(-((r - rad) / 5.0) ** 2)                                                     # [T,B,S,S]
    blobs = torch.zeros(1, B, S, S)
    for k in range(6):
        blobs += 0.5 * torch.exp(-((yy - by[:, k].view(1, B, 1, 1)) ** 2 + (xx - bx[:, k].view(1, B, 1, 1)) ** 2)
                                 / (2 * bs[:, k].view(1, B, 1, 1) ** 2))
    img = img + blobs + 0.05 * torch.randn(T, B, S, S, generator=g)
    img = (img - img.mean(dim=(2, 3), keepdim=True)) / img.std(dim=(2, 3), unbiased=False, keepdim=True)
    return img.view(T, B, 1, S, S).contiguous()


def ring_labels(B, S):
    yy, xx = np.mgrid[:S, :S]
    rad = np.sqrt((yy - S / 2) ** 2 + (xx - S / 2) ** 2)
    lab = np.zeros((S, S), np.uint8)
    for k, r in enumerate((52, 44, 36), start=1):
        lab[rad < r] = k
    return torch.from_numpy(lab)[None].repeat(B, 1, 1).contiguous()


def flow_net_kwargs(variant):
    if variant == "raft":      # SegFlowGaussian.py:875-969 with raft_config.yaml:41 raft_iters 12
        return dict(image_size=256, motion_appearance=False, dim_feedforward=2048, raft=True, raft_iters=12)
    ma = variant == "raft_config"
    return dict(image_size=256, motion_appearance=ma, dim_feedforward=3072 if ma else 2048)


def make_nets(variant, with_seg=True):
    from cineflow.models import SegFlowGaussian, Generic_UNet, OpticalFlowModelSuccessive, ModelWrap
    if variant == "successive":   # successive.yaml: in [6,128,256], out [64,128,256], d_model 512, 8 heads, PatchMerging downsampling (nnMTLTrainerV2FlowSuccessive.py:490-496)
        from cineflow.predict import ModelWrapFlow
        nets = [ModelWrapFlow(ModelWrap(OpticalFlowModelSuccessive(256, 1), OpticalFlowModelSuccessive(256, 6)))]
    else:
        nets = [SegFlowGaussian(**flow_net_kwargs(variant))]
    if with_seg:
        nets.append(Generic_UNet(1, 32, 4, 6))
    return nets


def load_nets(nets, dev, seed, world, rank):
    """rank 0 "reads the checkpoint" (the seeded fill); the other ranks receive it as ONE flat broadcast"""
    from cineflow.weights import seeded_state_dict
    from cineflow import parallel
    for i, net in enumerate(nets):
        shapes = {k: v for k, v in net.state_shapes().items() if not k.endswith("grid")}
        sd = seeded_state_dict(shapes, seed + i) if rank == 0 else None
        sd = parallel.broadcast_state_dict(sd, shapes, dev) if world > 1 else {k: v.to(dev) for k, v in sd.items()}
        if world > 1 and parallel.LAST_BROADCAST:
            log("rank %d: weight broadcast %d of %.1f MB took %.1f ms" % (rank, i, parallel.LAST_BROADCAST["bytes"] / 1e6, parallel.LAST_BROADCAST["seconds"] * 1e3))
        if dev.type == "cuda":
            net.load_state_dict(sd, dev)
        else:
            net._dry_checksum = float(sum(float(v.double().sum()) for v in sd.values()))
    return nets


def run_step(fnet, snet, frames, ed_labels, seg_mixed_precision=False):
    """one step of BASELINE config 4 (tools/layer_profile.py times the same call)"""
    from cineflow.inference import predict_cine_slices
    return predict_cine_slices(fnet, snet, frames, ed_labels, seg_mixed_precision=seg_mixed_precision)


def seg_precision_parity(fnet, snet, frames, ed_labels):
    """--seg-precision f16: what the one-term segmentation path changes against the f32-class path on the SAME inputs (the first 4 slices of
    the bench batch): per-class Dice of the label maps (the north-star bar is 1e-3), fraction of differing voxels, max |softmax difference|;
    the flow and the propagated labels do not depend on the switch (the flow network stays f32-class)."""
    import numpy as np
    n = min(4, frames.shape[1])
    fr = frames[:, :n].contiguous()
    ed = None if ed_labels is None else ed_labels[:n].contiguous()
    a = run_step(fnet, snet, fr, ed, False)
    b = run_step(fnet, snet, fr, ed, True)
    sa, sb = a["seg"].cpu().numpy(), b["seg"].cpu().numpy()
    dice = []
    for k in range(4):
        ta, tb = sa == k, sb == k
        den = ta.sum() + tb.sum()
        dice.append(None if den == 0 else round(float(2.0 * (ta & tb).sum() / den), 6))
    return {"dice_vs_f32_class_path": dice, "min_dice": min(d for d in dice if d is not None),
            "voxels_differing": round(float((sa != sb).mean()), 7), "max_abs_softmax_diff": round(float((a["softmax"] - b["softmax"]).abs().max()), 5),
            "max_abs_flow_diff_px": round(float((a["flow"] - b["flow"]).abs().max()), 7),      # run-to-run noise of the statistics atomics: the flow network is untouched
            "dice_bar_1e-3_holds": bool(min(d for d in dice if d is not None) >= 0.999), "slices": n, "frames": int(frames.shape[0])}


def cpu_baseline(variant, seed, T_sample):
    """The oracle (CPU PyTorch restatement pinned to the reference, oracle/) on a bounded sample of the same workload:
    one slice, T_sample frames, same architectures and seeded weights, all host cores."""
    from oracle import models as OM
    from cineflow.weights import fill_module_
    from cineflow.inference import chunk_orders
    from oracle import ops as OO
    torch.set_num_threads(host_threads())
    if variant == "warp":
        from oracle import metrics as OMet
        n = 512
        flow, img, lab = synthetic_pairs(n, 256, 1234)
        with torch.no_grad():
            OO.warp_bilinear(flow[:2], img[:2])  # warm-up
            t0 = time.perf_counter()
            OO.warp_bilinear(flow, img)
            OO.warp_labels(flow[None], lab[:, None].float())
            for b in range(n):
                OMet.jacobian_determinant(flow[b].permute(1, 2, 0).numpy())
            dt = time.perf_counter() - t0
        return {"value": n / dt, "unit": "frame pairs/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": "%d frame pairs of the same workload (oracle/: grid_sample warp, one-hot label warp, np.gradient Jacobian), %.1f s" % (n, dt)}
    if variant == "successive":
        wrap = fill_module_(OM.ModelWrap(OM.OpticalFlowModelSuccessive(256, 1), OM.OpticalFlowModelSuccessive(256, 6)), seed)
        fnet = lambda x: {"backward_flow": (wrap(x)[1]["cumulated"] if x.shape[0] > 2 else wrap(x)[1]["flow"][None])}   # noqa: E731
    else:
        fnet = fill_module_(OM.SegFlowGaussian(**flow_net_kwargs(variant)), seed)
    frames = synthetic_cine(1, T_sample, 256, 1234)
    with torch.no_grad():
        if variant == "raft":
            fnet(frames[:2])  # warm-up
            t0 = time.perf_counter()
            fnet(frames[:2])
            dt = time.perf_counter() - t0
            return {"value": 1.0 / dt, "unit": "frame pairs/s", "cores": torch.get_num_threads(), "kind": "port",
                    "sample": "1 frame pair x 12 RAFT iterations (oracle/ CPU PyTorch fp32), %.1f s" % dt}
        snet = fill_module_(OM.GenericUNet2D(1, 32, 4, 6), seed + 1)
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
RANK = int(os.environ.get("RANK", "0"))


def host_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))  # a 1-GPU box gives this process a 16-core share


# --------------------------------------------------------------------------------------------------- self-launch
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """Start n ranks of this script (one per GPU), relay rank 0's stdout (the JSON line) and return the worst exit code.
    Runs before this process has made any GPU call: the children are fresh processes, nothing is re-exec'd."""
    import threading
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    pending = set(range(n))
    try:
        while pending:
            time.sleep(0.2)
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    log("rank %d exited with code %d: stopping the other ranks" % (r, code))
                    for q in pending:
                        procs[q].terminate()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    reader.join(timeout=10)
    # only the JSON line goes to stdout; anything else rank 0 printed there (gloo / RCCL banners) is passed on through stderr
    for ln in b"".join(o for o in out0 if o).decode(errors="replace").splitlines():
        (sys.stdout if ln.lstrip().startswith("{") else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    return rc


# --------------------------------------------------------------------------------------------------- measurement
def read_profile(h, kid):
    from cineflow import _lib
    ms, work, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_long()
    _lib.check(h.cf_profile_read(kid, ctypes.byref(ms), ctypes.byref(work), ctypes.byref(n)), "cf_profile_read")
    return ms.value, work.value, n.value


def hbm_roofline(name, rec, pmc=None):
    ms, byts, n = rec
    if not n:
        return None
    ach = byts / (ms * 1e-3) / 1e9
    r = {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
         "traffic": None, "launches": n, "avg_launch_us": round(ms * 1e3 / n, 2)}
    if pmc in PMC_TRAFFIC_RATIO:      # bytes per launch: algorithmic x the PMC-measured ratio (a constant with its source, see PMC_TRAFFIC_RATIO)
        ratio, src = PMC_TRAFFIC_RATIO[pmc]
        r.update(traffic=round(byts / n * ratio), traffic_unit="HBM bytes per launch", algorithmic_bytes_per_launch=round(byts / n),
                 traffic_over_algorithmic=ratio, traffic_source=src)
    return r


def conv_roofline(h, dt):
    """roofline of the dominant convolution kernel from the per-launch event pairs recorded during the timed steps"""
    conv = [read_profile(h, k) for k in (0, 1, 2, 6)]
    stream = read_profile(h, 15)          # conv_stream_kernel: the persistent variant of the same 3-term algorithm for the 32 / 64-channel layers
    wino = read_profile(h, 16)            # conv_wino_kernel(s): row Winograd F(2,3) for the >= 128-channel 3x3 / stride 1 layers; work = direct-form flops
    one_tile = conv[3]
    conv[3] = tuple(a + b + c for a, b, c in zip(conv[3], stream, wino))
    dom = max(range(4), key=lambda i: conv[i][0])
    ms, flops, n = conv[dom]
    if not n:
        return None
    ach = flops / (ms * 1e-3) / 1e12
    if dom == 3:
        # the f16 hi/lo-split kernels issue 3 f16 MFMAs per MAC they execute: priced against the dense f16 MFMA peak the direct form's
        # ceiling is 1/3.  `achieved` counts ALGORITHMIC (direct-form) flops for every kernel of the family; the Winograd kernel executes
        # 2/3 of them (12 instead of 18 k-steps per pair of output columns), so `mfma_issue_frac` -- the fraction of the f16 MFMA peak the
        # ISSUED MFMAs reach -- takes its flops x 2/3.
        issued = 3.0 * (one_tile[1] + stream[1] + wino[1] * 2.0 / 3.0) / (ms * 1e-3) / 1e12
        per = {nm: {"TFLOP/s": round(r[1] / (r[0] * 1e-3) / 1e12, 1), "launches": r[2], "avg_launch_us": round(r[0] * 1e3 / r[2], 2),
                    "share_of_step_time": round(r[0] * 1e-3 / dt, 3)}
               for nm, r in (("conv_f16s_kernel", one_tile), ("conv_stream_kernel", stream), ("conv_wino_kernel", wino)) if r[2]}
        return {"bound": "mfma", "kernel": "conv_f16s_kernel + conv_stream_kernel + conv_wino_kernel (f16 MFMA, 3-term hi/lo split, fp32 accumulate; one tile per "
                                           "workgroup / persistent for 32- and 64-channel layers / row Winograd F(2,3) for the >= 128-channel 3x3 layers)",
                "per_kernel": per, "achieved": round(ach, 3),
                "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_F16_PEAK_TFLOPS, 4),
                "flops_counted": "algorithmic (direct-form) flops of every layer, also for the Winograd kernel",
                "mfma_issue_frac": round(issued / MFMA_F16_PEAK_TFLOPS, 4), "vs_fp32_mfma_peak": round(ach / MFMA_F32_PEAK_TFLOPS, 3),
                "sustained_mfma_peak_measured": MFMA_F16_SUSTAINED_TFLOPS, "mfma_issue_frac_of_sustained": round(issued / MFMA_F16_SUSTAINED_TFLOPS, 4),
                "mfma_busy": MFMA_BUSY,
                "traffic": None, "traffic_over_algorithmic": PMC_TRAFFIC_RATIO["conv_f16s"][0], "traffic_source": PMC_TRAFFIC_RATIO["conv_f16s"][1],
                "traffic_note": "HBM bytes per launch = algorithmic bytes (inputs read once + outputs written once, per layer shape) x traffic_over_algorithmic; "
                                "the kernel is MFMA-bound, so the live line prices flops and carries the PMC byte ratio as a constant",
                "launches": n, "avg_launch_us": round(ms * 1e3 / n, 2), "share_of_step_time": round(ms * 1e-3 / dt, 3)}
    return {"bound": "mfma", "kernel": "conv_igemm_f32_kernel<%d,2>" % (1, 2, 4)[dom], "achieved": round(ach, 3),
            "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
            "launches": n, "avg_launch_us": round(ms * 1e3 / n, 2), "share_of_step_time": round(ms * 1e-3 / dt, 3)}


def timed(fn, steps, warmup, h, dev, what):
    """W untimed steps, then exactly K steps between barrier + synchronize pairs; returns the MAX over ranks"""
    from cineflow import parallel, _lib
    for _ in range(warmup):
        fn()
        torch.cuda.synchronize()
        log("%s: warm-up step done" % what)
    if RANK == 0:      # per-launch event pairs (the roofline's source) on rank 0 only: the other ranks of an N-GPU run just do the work
        _lib.check(h.cf_profile_enable(60000), "cf_profile_enable")
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0
    parallel.barrier()
    dt = time.perf_counter() - t0
    del out
    dt = parallel.max_over_ranks(dt, dev)
    log("%s: rank %d: %d steps in %.3f s (max over ranks incl. barrier: %.3f s)" % (what, RANK, steps, mine, dt))
    return dt


def bench_joint(args, dev, h, world, rank):
    """BASELINE config 4 (the headline metric)"""
    from cineflow.inference import predict_cine_slices
    B, T, S = args.slices, args.frames, 256
    fnet, snet = load_nets(make_nets(args.variant), dev, 1234, world, rank)
    # patient i of the global list is seeded 1234+i and handled by rank i % world
    frames = synthetic_cine(B, T, S, 1234 + rank).to(dev)
    ed_labels = ring_labels(B, S).to(dev)
    log("joint: weights + inputs resident (rank %d/%d, B=%d, T=%d)" % (rank, world, B, T))
    mp = args.seg_precision == "f16"
    parity = seg_precision_parity(fnet, snet, frames, ed_labels) if (mp and rank == 0) else None
    dt = timed(lambda: run_step(fnet, snet, frames, ed_labels, mp), args.steps, args.warmup, h, dev, "joint")
    roofline = conv_roofline(h, dt)
    corr = [read_profile(h, k) for k in (3, 4, 5)]
    tot = tuple(sum(c[i] for c in corr) for i in range(3))
    roofline_corr = hbm_roofline("corr_volume_mfma_kernel<1|2|4> (persistent; 2-D banded f16-MFMA products of hi/lo-split operands, fp32 accumulate)", tot, pmc="corr")
    if roofline_corr:
        roofline_corr["per_level"] = {"s%d" % s: {"GB/s": round(c[1] / (c[0] * 1e-3) / 1e9, 1), "avg_launch_us": round(c[0] * 1e3 / c[2], 2)}
                                      for s, c in zip((1, 2, 4), corr) if c[2]}
    h.cf_profile_enable(0)
    frames_total = world * B * T * args.steps
    return {
        "metric": "cine frames/sec (seg+flow) at 256x256",
        "value": round(frames_total / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f16 MFMA, f32 accumulate: flow network hi/lo split (f32-class), segmentation U-Net one-term (fp16 operands)" if mp else
                  "f16 MFMA hi/lo split, f32 accumulate (f32-class)") if args.conv_mode == "f16s" else "f32", "data": "synthetic",
        "config": {"workload": "BASELINE config 4: joint seg+flow over 256x256x%d cine slices; Generic_UNet(32 base, 6 pools) 4-flip TTA on "
                               "every frame + %s two-chunk ED-anchored recurrence + fused label warp"
                               % (T, "ModelWrap(OpticalFlowModelSuccessive x 2, successive.yaml: d_model 512, 8 heads)" if args.variant == "successive"
                                  else "SegFlowGaussian(%s.yaml)" % args.variant),
                   "slices_per_step_per_gpu": B, "frames_per_slice": T, "image": "256x256", "sharding": "patients, rank = part_id"},
        "roofline": roofline, "roofline_corr": roofline_corr,
        **({"seg_precision": "f16: the segmentation U-Net's convolutions in the one-term product mode (operands rounded to fp16, fp32 accumulate / norms) = "
                             "the reference's mixed_precision=True on that path; the flow network stays f32-class.  NOT the headline configuration "
                             "(default: --seg-precision f32)", "seg_precision_parity": parity} if mp else {}),
    }


def bench_raft(args, dev, h, world, rank, steps, warmup):
    """BASELINE config 3: RAFT flow, 12 update iterations per 256x256 frame pair (fmaps [B,256,32,32]); a step = B pairs"""
    B, S = args.pairs, 256
    (fnet,) = load_nets(make_nets("raft", with_seg=False), dev, 1234, world, rank)
    frames = synthetic_cine(B, 2, S, 4321 + rank).to(dev)
    log("raft: weights + inputs resident (rank %d/%d, B=%d pairs)" % (rank, world, B))
    dt = timed(lambda: fnet(frames)["backward_flow"], steps, warmup, h, dev, "raft")
    roofline = conv_roofline(h, dt)
    allp, look, up = (read_profile(h, k) for k in (7, 8, 9))
    h.cf_profile_enable(0)
    pairs = world * B * steps
    return {
        "metric": "RAFT frame pairs/sec (12 update iterations) at 256x256", "value": round(pairs / dt, 2), "unit": "frame pairs/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 2), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f16 MFMA hi/lo split, f32 accumulate (f32-class)" if args.conv_mode == "f16s" else "f32",
        "data": "synthetic",
        "config": {"workload": "BASELINE config 3: SegFlowGaussian(raft=True, raft_iters=12) on 256x256 frame pairs: two encoders, all-pairs "
                               "correlation pyramid (fmaps [B,256,32,32], 4 levels), 12 x (lookup r=4 -> motion encoder -> SepConvGRU -> flow / mask "
                               "heads -> convex upsampling), memory-encoder refresh", "pairs_per_step_per_gpu": B, "image": "256x256"},
        "roofline": roofline,
        "roofline_allpairs": hbm_roofline("allpairs_pyramid_kernel (f16-split MFMA product, the 4 pyramid levels from one set of accumulators); 7.67 MB / pair", allp),
        "roofline_lookup": hbm_roofline("corr_lookup_tiled_kernel (324 channels written once per iteration; gathers are cache traffic)", look),
        "roofline_upsample": hbm_roofline("convex_upsample_rows_kernel (576-channel mask read once per iteration, eight adjacent outputs per thread)", up),
    }


def synthetic_pairs(B, S, seed):
    """B frame pairs' worth of VoxelMorph inputs: a smooth displacement field (coarse 16x16 Gaussian field of sigma 3 px, bilinearly
    upsampled -- cardiac motion is a few pixels and smooth), a moving image and an ED label map with the four classes as nested discs"""
    g = torch.Generator().manual_seed(seed)
    coarse = 3.0 * torch.randn(B, 2, 16, 16, generator=g)
    flow = torch.nn.functional.interpolate(coarse, size=(S, S), mode="bilinear", align_corners=True).contiguous()
    img = synthetic_cine(B, 1, S, seed + 1)[0].reshape(B, 1, S, S).contiguous()
    yy, xx = torch.meshgrid(torch.arange(S, dtype=torch.float32), torch.arange(S, dtype=torch.float32), indexing="ij")
    c = S / 2 + 8.0 * torch.randn(B, 2, generator=g)
    r = torch.sqrt((yy[None] - c[:, 0, None, None]) ** 2 + (xx[None] - c[:, 1, None, None]) ** 2)
    lab = ((r < 0.30 * S).to(torch.uint8) + (r < 0.20 * S).to(torch.uint8) + (r < 0.12 * S).to(torch.uint8)).contiguous()
    return flow, img, lab


def bench_warp(args, dev, h, world, rank, steps, warmup):
    """BASELINE config 2: the VoxelMorph warp of 256x256 frame pairs as voxelmorph_saver_* / compute_jacobian use it; a step = B pairs:
    SpatialTransformer on the moving image, warp_linear label propagation (one-hot -> warp -> argmax), Jacobian determinant"""
    from cineflow import ops
    B, S = args.pairs, 256
    flow, img, lab = (t.to(dev) for t in synthetic_pairs(B, S, 4321 + rank))
    fl5 = flow[None]
    log("warp: inputs resident (rank %d/%d, B=%d pairs)" % (rank, world, B))

    def step():
        return ops.warp_bilinear(flow, img), ops.warp_labels(fl5, lab), ops.jacobian_det(flow)

    dt = timed(step, steps, warmup, h, dev, "warp")
    recs = [read_profile(h, k) for k in (11, 12, 13)]
    h.cf_profile_enable(0)
    names = ["warp_bilinear_2d_v4_kernel (flow 8 B + image 4 B read, 4 B written per pixel)",
             "warp_labels_2d_v4_kernel (flow 8 B + label 1 B read, 1 B written per pixel)",
             "jacobian_det_2d_v4_kernel (displacement 8 B read, float64 determinant 8 B written per pixel)"]
    roofs = [hbm_roofline(n, r) for n, r in zip(names, recs)]
    dom = max(range(3), key=lambda i: recs[i][0])
    pairs = world * B * steps
    return {
        "metric": "VoxelMorph warp frame pairs/sec at 256x256", "value": round(pairs / dt, 1), "unit": "frame pairs/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 (Jacobian f64)", "data": "synthetic",
        "config": {"workload": "BASELINE config 2: per 256x256 frame pair, SpatialTransformer warp of the moving image, warp_linear label "
                               "propagation (4 classes) and the Jacobian determinant of the displacement", "pairs_per_step_per_gpu": B,
                   "image": "256x256"},
        "roofline": roofs[dom], "roofline_other": [r for i, r in enumerate(roofs) if i != dom],
    }


def bench_api(args, dev, world, rank):
    """The file-level API end to end (reference predict.py:665-780 -> :228-354 -> :1008-1110): `predict_from_folder` on
    P synthetic patients (8 slices x T frames x 256x256 each, one gzip NIfTI per frame) in a tmpfs folder: model folder read from disk,
    NIfTI read + crop-to-nonzero + z-score, cropped slices of several patients batched per device launch, NIfTI / NPZ export in the background.
    A step = one predict_from_folder call over all P patients; `value` = frames/s including all of that (the model itself is loaded by the
    untimed warm-up call and stays cached, as a serving process would keep it)."""
    from cineflow import predict as P
    from cineflow import parallel
    from cineflow.nifti import write_nifti
    from cineflow.weights import seeded_state_dict
    npat, Z, T, S = args.patients, 8, args.frames, 256
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    root = tempfile.mkdtemp(prefix="cineflow_api_r%d_" % rank, dir=base)
    try:
        t0 = time.perf_counter()
        fnet, snet = make_nets("video")
        plans = P.default_plans(image_size=S, crop_size=S, flow_variant="video", seg_base=32, seg_pool=6)
        model = os.path.join(root, "model")
        P.save_model_folder(model, snet, fnet, plans, fold=0, seg_sd=seeded_state_dict(snet.state_shapes(), 1235),
                            flow_sd=seeded_state_dict({k: v for k, v in fnet.state_shapes().items() if not k.endswith("grid")}, 1234))
        del fnet, snet
        inp = os.path.join(root, "in")
        for p in range(npat):
            frames = synthetic_cine(Z, T, S, 1234 + rank * npat + p)[:, :, 0].numpy() * 60.0 + 200.0      # [T,Z,S,S], scanner-like intensities
            pat = "patient%03d" % p
            os.makedirs(os.path.join(inp, pat))
            for t in range(T):
                write_nifti(os.path.join(inp, pat, "%s_frame%02d_0000.nii.gz" % (pat, t)), frames[t].astype(np.float32), (1.25, 1.25, 10.0), (0.0, 0.0, 0.0))
        log("api: model folder + %d patients x %d frames written to %s in %.1f s" % (npat, T, root, time.perf_counter() - t0))
        threads = host_threads()
        # pool sizes of the call (the reference's num_threads_preprocessing / num_threads_nifti_save); the calling thread needs a core of
        # its own for the launch stream, so the two pools share the rest
        pre_threads = args.pre_threads or max(1, threads // 4)                       # measured on a 16-thread box: 4 + 10 > 6 + 8 > 8 + 6 (reading is light since the
        save_threads = args.save_threads or max(1, threads - pre_threads - 2)        # case makes one device round trip; the export compresses three files per frame)

        def call(tag, n):
            out = os.path.join(root, "out_" + tag)
            sub = os.path.join(root, "in_" + tag)
            os.makedirs(sub)
            for p in range(n):
                os.symlink(os.path.join(inp, "patient%03d" % p), os.path.join(sub, "patient%03d" % p))
            P.predict_from_folder(model, sub, out, [0], False, pre_threads, save_threads, None, 0, 1, True, disable_postprocessing=True)
            tim = dict(P.LAST_TIMING)
            shutil.rmtree(out)
            return tim

        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):            # the API prints per-patient progress like the reference; keep stdout for the JSON line
            for w in range(max(1, args.warmup)):
                tim = call("warm%d" % w, min(2, npat))                 # loads the model folder (cached afterwards), warms allocator and kernels
                log("api: warm-up call done (model load %.1f s)" % tim["load_s"])
            load_s = tim["load_s"]
            parallel.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tims = [call("step%d" % k, npat) for k in range(args.steps)]
            torch.cuda.synchronize()
            parallel.barrier()
            dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
        split = {k: round(sum(t[k] for t in tims) / args.steps, 3) for k in ("preprocess_wait_s", "preprocess_work_s", "device_s", "export_wait_s", "export_work_s", "total_s")}
        for k in ("device_prepare_s", "device_networks_s", "device_finish_s"):       # CF_API_PROFILE=1 only
            if k in tims[0]:
                split[k] = round(sum(t[k] for t in tims) / args.steps, 3)
        split["device_batches_per_step"] = tims[0]["device_batches"]
        split["model_load_s_cold"] = round(load_s, 3)
        frames_total = world * npat * Z * T * args.steps
        return {
            "metric": "cine frames/sec (seg+flow) at 256x256, predict_from_folder end to end", "value": round(frames_total / dt, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup), "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16 MFMA hi/lo split, f32 accumulate (f32-class)", "data": "synthetic",
            "config": {"workload": "predict_from_folder (nnunet/inference/predict.py:665-780 API) on %d synthetic patients x %d slices x %d frames x 256x256 "
                                   "from gzip NIfTI files in tmpfs: read + preprocess, Generic_UNet 4-flip TTA + SegFlowGaussian(video.yaml) + label warp with "
                                   "up to %d slices of several patients per device batch, NIfTI / NPZ export; host work INSIDE the timed region" % (npat, Z, T, P.MAX_SLICES_PER_LAUNCH),
                       "patients_per_step_per_gpu": npat, "slices_per_patient": Z, "frames_per_slice": T, "host_threads": threads,
                       "num_threads_preprocessing": pre_threads, "num_threads_nifti_save": save_threads},
            "wall_time_split_per_step": split,
            "note": "preprocess_wait / device / export_wait are what the calling thread spent blocked in each stage (they add up to total); *_work are summed "
                    "thread times of the background pools; the device-only rate of the same networks is the default variant's `value`",
        }
    finally:
        shutil.rmtree(root, ignore_errors=True)


def dry_run(args, world, rank):
    """gloo + CPU tensors: launch, rendezvous, flat weight broadcast, shard seeding, barrier and max-over-ranks exactly as in
    the measured path; the GPU step is a sleep.  No libcineflow_hip.so call (there is no GPU to make one on)."""
    from cineflow import parallel
    dev = torch.device("cpu")
    variant = "video" if args.variant == "api" else args.variant
    nets = [] if variant == "warp" else load_nets(make_nets(variant, with_seg=variant != "raft"), dev, 1234, world, rank)
    sums = [n._dry_checksum for n in nets]
    frames = synthetic_cine(2, 4, 64, 1234 + rank)
    parallel.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (1 + rank))
    parallel.barrier()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    same = all(abs(parallel.max_over_ranks(s, dev) + parallel.max_over_ranks(-s, dev)) < 1e-9 for s in sums)
    total = parallel.sum_over_ranks(float(frames.numel()), dev)
    if rank == 0:
        print(json.dumps({"metric": "cine frames/sec (seg+flow) at 256x256", "value": None, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(dt / max(1, args.steps) * 1e3, 2), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none (dry run)", "data": "synthetic", "dry_run": True,
                          "config": {"workload": "dry run of the multi-rank path (gloo, CPU tensors, no GPU work)", "variant": args.variant},
                          "weights_identical_on_all_ranks": bool(same), "ranks_reporting": int(round(total / frames.numel()))}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--slices", type=int, default=None, help="cine slices per step and per GPU (B); 32 -> 48 -> 64 measured 707.8 -> 713.0 -> 717.4 frames/s on one box")
    ap.add_argument("--frames", type=int, default=30, help="frames per cine slice (T)")
    ap.add_argument("--pairs", type=int, default=None, help="frame pairs per step and per GPU (default: 128 for raft, 960 for warp)")
    ap.add_argument("--patients", type=int, default=16, help="api variant: synthetic patients (8 slices each) per step and per GPU")
    ap.add_argument("--pre-threads", type=int, default=0, help="api variant: num_threads_preprocessing (default: a quarter of the host threads)")
    ap.add_argument("--save-threads", type=int, default=0, help="api variant: num_threads_nifti_save (default: the rest minus two)")
    ap.add_argument("--variant", default="video", choices=["video", "raft_config", "successive", "raft", "warp", "api"],
                    help="video / raft_config: BASELINE config 4 with that flow dispatch; raft: BASELINE config 3 as the headline line; "
                         "warp: BASELINE config 2 (VoxelMorph warp of frame pairs); successive: config 4 with the successive.yaml pair of networks; "
                         "api: predict_from_folder end to end on synthetic patients")
    ap.add_argument("--no-raft", action="store_true", help="skip the nested BASELINE config 3 measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--conv-mode", default="f16s", choices=["f16s", "f32"], help="f16s: f16-MFMA hi/lo split (default); f32: exact fp32 MFMA")
    ap.add_argument("--seg-precision", default="f32", choices=["f32", "f16"],
                    help="f32 (default, the headline): every network f32-class; f16: the segmentation U-Net in the one-term fp16 product mode "
                         "(the reference's mixed_precision=True on the segmentation path), reported as a second, labelled line with its parity figures")
    ap.add_argument("--cpu-frames", type=int, default=30, help="frames of the one-slice CPU baseline sample (30 = one whole cine slice, ~15 s on 16 cores)")
    ap.add_argument("--dry-run", action="store_true", help="exercise the multi-rank path on gloo / CPU tensors without GPU work")
    args = ap.parse_args()
    if args.pairs is None:
        args.pairs = 960 if args.variant == "warp" else 128      # RAFT: 64 / 128 / 256 pairs per step measured 995 / 1033 / 1036 pairs/s
    if args.slices is None:
        # successive.yaml's model1 takes every frame of a half sequence through its encoder as ONE batch (T x B images): 16 slices already
        # give 256-image launches
        args.slices = 16 if args.variant == "successive" else 64

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))           # before any GPU call in this process

    from cineflow import parallel
    if args.dry_run:
        rank, world, _ = parallel.init_from_env(backend="gloo")
        assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
        dry_run(args, world, rank)
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    from cineflow import _lib
    rank, world, local_rank = parallel.init_from_env()
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs a GPU (cineflow has no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    h = _lib.lib()
    from cineflow import ops as _ops
    _ops.set_conv_mode(args.conv_mode)

    if args.variant == "raft":
        line = bench_raft(args, dev, h, world, rank, args.steps, args.warmup)
    elif args.variant == "warp":
        line = bench_warp(args, dev, h, world, rank, args.steps, args.warmup)
    elif args.variant == "api":
        line = bench_api(args, dev, world, rank)
        args.no_cpu_baseline = True       # the CPU path of this metric is the default variant's cpu_baseline (the oracle has no file-level API)
    else:
        line = bench_joint(args, dev, h, world, rank)
        if not args.no_raft and args.variant != "successive":
            torch.cuda.empty_cache()
            line["config3_raft"] = bench_raft(args, dev, h, world, rank, max(1, args.steps), 1)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            log("timing the CPU baseline (oracle, %d threads) ..." % host_threads())
            line["cpu_baseline"] = cpu_baseline(args.variant, 1234, args.cpu_frames)
            log("CPU baseline done")
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
